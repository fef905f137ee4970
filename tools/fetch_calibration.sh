#!/bin/bash
# ON THE GPU BOX: two PMC passes over tools/fetch_calibration.py -> $1 (JSON)
out="${1:-/root/repo/gpurun_out/fetch_calibration.json}"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  for attempt in 1 2 3 4; do
    rm -rf /tmp/fc && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/fc -o c -- python /root/repo/tools/fetch_calibration.py > /dev/null 2>&1
    f="$(find /tmp/fc -name '*counter_collection.csv' 2>/dev/null | head -1)"
    if [ -n "$f" ]; then python /root/repo/tools/pmc_summary.py "$f" $c > /tmp/fc_$c.json && break; fi
  done
done
python - "$out" <<'PY'
import json, sys
n = 12599920
f = json.load(open("/tmp/fc_FETCH_SIZE.json")).get("k_adamw", {})
w = json.load(open("/tmp/fc_WRITE_SIZE.json")).get("k_adamw", {})
exp_r, exp_w = 16.0 * n, 18.0 * n
res = {"_what": "stand-alone k_adamw sweep of a 12,599,920-entry fp32 vector under rocprofv3 --pmc (separate passes, kernel trace "
                "only): known 16 B read / 18 B written per parameter (tools/fetch_calibration.py)",
       "expected_read_MB": exp_r / 1e6, "expected_write_MB": exp_w / 1e6,
       "FETCH_SIZE_KiB_per_dispatch": f.get("avg"), "WRITE_SIZE_KiB_per_dispatch": w.get("avg"), "dispatches": f.get("dispatches"),
       "fetch_measured_over_expected": (f["avg"] * 1024 / exp_r) if f.get("avg") else None,
       "write_measured_over_expected": (w["avg"] * 1024 / exp_w) if w.get("avg") else None}
if res["fetch_measured_over_expected"]:
    res["read_side_multiplier"] = 1.0 / res["fetch_measured_over_expected"]
json.dump(res, open(sys.argv[1], "w"), indent=1)
print(json.dumps(res))
PY
