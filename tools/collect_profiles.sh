#!/bin/bash
# Run ON THE GPU BOX (through gpurun): final bench line, rocprofv3 kernel summary + timeline, PMC traffic passes.
# Writes small summaries under gpurun_out/$1/ ; copy what should be judged into profiles/.
set -u
tag="${1:-r01_final}"; out="/root/repo/gpurun_out/$tag"; mkdir -p "$out"
cd /root/repo
python bench.py > "$out/bench.json" 2> "$out/bench.stderr"; tail -c 400 "$out/bench.json"; echo
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o k -- python /root/repo/bench.py --steps 100 --warmup 300 --no-cpu-baseline > "$out/bench_under_rocprof.json" 2>/dev/null
cp "$(find /tmp/pk -name '*kernel_stats.csv' | head -1)" "$out/kernel_stats.csv"
python /root/repo/tools/trace_tail.py "$(find /tmp/pk -name '*kernel_trace.csv' | head -1)" "$out/timeline_tail.csv" 2400
# PMC passes (separate --pmc runs, kernel trace only): once in the default regime and once with the short command line the
# round-end driver has used (--steps 20 --warmup 5), so that bench.py finds measured traffic for either
for regime in "300 200" "5 20"; do
  set -- $regime; w=$1; st=$2; rd="$out/pmc_w${w}_s${st}"; mkdir -p "$rd"
  for c in FETCH_SIZE WRITE_SIZE; do
    for attempt in 1 2 3 4 5; do  # rocprofv3 --pmc occasionally segfaults at exit on this image: retry, the passes are independent
      rm -rf /tmp/pc && NSR_BENCH_REGIME_OUT="$rd/bench_regime.json" rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pc -o c -- python /root/repo/bench.py --steps $st --warmup $w --no-cpu-baseline > /dev/null 2>&1
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
ls -la "$out"
