#!/bin/bash
# the PMC passes of tools/collect_profiles_r05.sh alone (lean bench command lines WITHOUT the whole-run / forms A/B blocks, so that
# the dispatch window of the roofline's 64 instrumented steps is what the counters are averaged over)
set -u
out=/root/repo/gpurun_out/r05_final; mkdir -p "$out"
LEAN="--no-cpu-baseline --no-other-workloads --no-boundary-path --no-whole-run"
export NSR_BENCH_NO_STEADY=1 NSR_BENCH_NO_FORMS_AB=1
cd /tmp && export TMPDIR=/tmp
for regime in "20 200" "5 20"; do
  set -- $regime; w=$1; st=$2; rd="$out/pmc_w${w}_s${st}"; mkdir -p "$rd"; rm -f "$rd"/*
  cp "$out/fetch_calibration.json" "$rd/" 2>/dev/null
  for c in FETCH_SIZE WRITE_SIZE; do
    for attempt in 1 2 3 4; do
      rm -rf /tmp/pc && NSR_BENCH_REGIME_OUT="$rd/bench_regime.json" rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pc -o c -- python /root/repo/bench.py --steps $st --warmup $w $LEAN > /dev/null 2>&1
      f="$(find /tmp/pc -name '*counter_collection.csv' 2>/dev/null | head -1)"
      if [ -n "$f" ]; then python /root/repo/tools/pmc_summary.py "$f" $c "$rd/bench_regime.json" > "$rd/pmc_$c.json" && break; fi
    done
  done
  python /root/repo/tools/pmc_traffic.py "$rd" "$rd/pmc_traffic.json"
done
python - "$out" <<'PY'
import json, os, sys
out = sys.argv[1]
regimes = {}
for d in sorted(os.listdir(out)):
    p = os.path.join(out, d, "pmc_traffic.json")
    if d.startswith("pmc_w") and os.path.exists(p):
        regimes[d[4:]] = json.load(open(p))
json.dump({"_what": "HBM-side traffic per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py, one entry per "
                    "(warmup, steps) command line; see tools/pmc_traffic.py", "regimes": regimes},
          open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
PY
python -c "
import json; p=json.load(open('$out/pmc_traffic.json'))
for k,v in p['regimes'].items(): print(k, v['hashgrid_backward_params']['bytes_per_launch']/1e6, v['hashgrid_backward_params']['samples_per_launch'], {a:b for a,b in v['_raw_KiB_per_dispatch'].items() if 'own' in a or 'slab' in a})"
