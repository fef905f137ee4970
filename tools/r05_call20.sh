#!/bin/bash
# ON THE GPU BOX: key 10 (table backward on the helper stream): parity test, then same-process A/B at two regimes
set -u
out=/root/repo/gpurun_out/r05s; mkdir -p "$out"
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q -k "helper_stream or forms or pipelined" 2>&1 | tail -6 > "$out/pytest.txt"; tail -4 "$out/pytest.txt"
for at in 700 2500; do
  NSR_VARIANTS=round5_forms,round5_table_on_helper,round5_table_on_helper_cap512 timeout 600 python tools/step_variants.py $at 160 4 > "$out/table_on_helper_$at.json" 2> "$out/table_on_helper_$at.err"
  python - "$out/table_on_helper_$at.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    for k, v in d["settings"].items():
        print(f"{k:32s} mean {v['mean_ms']:.4f} ms  {v['ms_per_step']}  kept {v['kept_per_step']}")
except Exception as e:
    print("failed", e)
PY
  tail -3 "$out/table_on_helper_$at.err"
done
