#!/bin/bash
set -u
out=/root/repo/gpurun_out/r05j; mkdir -p "$out"
cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_round5.py -q -k "lazy or halves or pipelined" 2>&1 | tail -4
timeout 900 python tools/step_variants.py 2500 160 4 > "$out/variants_2500.json" 2> "$out/variants_2500.err"; tail -3 "$out/variants_2500.err"
python - "$out/variants_2500.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    for k, v in d["settings"].items():
        print(f"{k:32s} mean {v['mean_ms']:.4f} ms  {v['ms_per_step']}  kept {v['kept_per_step']}")
except Exception as e:
    print("no variants json", e)
PY
NSR_VARIANTS=shipped_pipelined_encode bash tools/timeline_tail.sh "$out/timeline_pipelined.csv" 140 -- python /root/repo/tools/step_variants.py 700 64 1 > "$out/timeline_pipelined_summary.txt" 2>&1; head -12 "$out/timeline_pipelined_summary.txt"
