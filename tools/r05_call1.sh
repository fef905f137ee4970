#!/bin/bash
# round 5, first GPU call: parity of the new kernels, then the same-process A/B of the step's forms and a timeline of the new step
set -u
out=/root/repo/gpurun_out/r05a; mkdir -p "$out"
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q 2>&1 | tail -25 > "$out/pytest_round5.txt"; tail -12 "$out/pytest_round5.txt"
timeout 600 python tools/step_variants.py 700 160 3 > "$out/variants_steady.json" 2> "$out/variants_steady.err"; tail -3 "$out/variants_steady.err"
python - "$out/variants_steady.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    for k, v in d["settings"].items():
        print(f"{k:28s} mean {v['mean_ms']:.4f} ms  {v['ms_per_step']}  host {v['host_ms_per_step']}  kept {v['kept_per_step'][0]}")
except Exception as e:
    print("no variants json", e)
PY
NSR_VARIANTS=all_on bash tools/timeline_tail.sh "$out/timeline_all_on.csv" 140 -- python /root/repo/tools/step_variants.py 700 64 1 > "$out/timeline_all_on_summary.txt" 2>&1; head -30 "$out/timeline_all_on_summary.txt"
NSR_VARIANTS=all_off bash tools/timeline_tail.sh "$out/timeline_all_off.csv" 160 -- python /root/repo/tools/step_variants.py 700 64 1 > "$out/timeline_all_off_summary.txt" 2>&1; head -24 "$out/timeline_all_off_summary.txt"
