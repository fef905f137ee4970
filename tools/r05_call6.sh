#!/bin/bash
set -u
out=/root/repo/gpurun_out/r05g; mkdir -p "$out"
cd /root/repo
for n in 2500; do
timeout 900 python tools/step_variants.py $n 160 4 > "$out/variants_$n.json" 2> "$out/variants_$n.err"; tail -3 "$out/variants_$n.err"
python - "$out/variants_$n.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    for k, v in d["settings"].items():
        print(f"{k:32s} mean {v['mean_ms']:.4f} ms  {v['ms_per_step']}  kept {v['kept_per_step']}")
except Exception as e:
    print("no variants json", e)
PY
done
