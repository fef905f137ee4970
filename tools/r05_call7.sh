#!/bin/bash
set -u
out=/root/repo/gpurun_out/r05h; mkdir -p "$out"; rm -f "$out/grad_report.jsonl"
cd /root/repo
NSR_GRAD_REPORT="$out/grad_report.jsonl" NSR_GRAD_NO_ASSERT=1 timeout 1500 python -m pytest tests/test_gpu_golden.py tests/test_gpu_fused_neus.py tests/test_gpu_fused_neus_bg.py tests/test_gpu_models_entry.py tests/test_gpu_mlp.py tests/test_gpu_boundary_trace.py -q 2>&1 | tail -5
python - "$out/grad_report.jsonl" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1])]
bad = [r for r in rows if r["rel_l2"] > r["rel_bound"] or r["cosine"] < r["cos_bound"]]
print(len(rows), "measurements,", len(bad), "outside 1e-2 / 0.999")
for r in sorted(bad, key=lambda r: -r["rel_l2"]):
    print(f"{r['rel_l2']:.4f} cos {r['cosine']:.6f}  {r['test'].split('::')[-1][:60]}  {r['what']}")
print("max rel among passing", max((r["rel_l2"] for r in rows if r not in bad), default=None))
PY
