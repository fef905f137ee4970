#!/bin/bash
set -u
out=/root/repo/gpurun_out/r05c; mkdir -p "$out"
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q -k "async_trainer or step_with" 2>&1 | tail -8 > "$out/pytest_round5.txt"; tail -4 "$out/pytest_round5.txt"
timeout 900 python tools/step_variants.py 2500 160 4 > "$out/variants_2500.json" 2> "$out/variants_2500.err"; tail -3 "$out/variants_2500.err"
python - "$out/variants_2500.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    for k, v in d["settings"].items():
        print(f"{k:32s} mean {v['mean_ms']:.4f} ms  {v['ms_per_step']}  host {v['host_ms_per_step'][:2]}  kept {v['kept_per_step']}")
except Exception as e:
    print("no variants json", e)
PY
NSR_VARIANTS=shipped_late_wgrad bash tools/timeline_tail.sh "$out/timeline_late_wgrad.csv" 140 -- python /root/repo/tools/step_variants.py 700 64 1 > "$out/timeline_late_wgrad_summary.txt" 2>&1; head -24 "$out/timeline_late_wgrad_summary.txt"
NSR_VARIANTS=shipped bash tools/timeline_tail.sh "$out/timeline_shipped.csv" 140 -- python /root/repo/tools/step_variants.py 700 64 1 > "$out/timeline_shipped_summary.txt" 2>&1; head -24 "$out/timeline_shipped_summary.txt"
