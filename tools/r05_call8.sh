#!/bin/bash
set -u
out=/root/repo/gpurun_out/r05i; mkdir -p "$out"
cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_round5.py tests/test_gpu_models_entry.py tests/test_gpu_entry_protocol.py tests/test_gpu_nccl_single_rank.py -x -q 2>&1 | tail -12 > "$out/pytest.txt"; tail -6 "$out/pytest.txt"
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
for mode in lazy eager; do
  if [ $mode = eager ]; then export NSR_BOUNDARY_EAGER=1; else unset NSR_BOUNDARY_EAGER; fi
  python - > "$out/boundary_$mode.json" 2> "$out/boundary_$mode.err" <<'PY'
import json, sys, torch
sys.argv = ['bench.py']
sys.path.insert(0, '/root/repo')
import bench
print(json.dumps(bench.boundary_path(torch.device('cuda', 0))))
PY
  tail -c 700 "$out/boundary_$mode.json"; echo; tail -2 "$out/boundary_$mode.err"
done
