#!/bin/bash
# A/B on the GPU box: two builds of the table backward (NSR_HIP_LIB) -> bench lines (lean) in both regimes + FETCH_SIZE of the
# owner kernel.  usage: tools/ab_list.sh <tag> <lib> <lib> ...
set -u
tag="$1"; shift; out="/root/repo/gpurun_out/$tag"; mkdir -p "$out"
LEAN="--no-cpu-baseline --no-other-workloads --no-boundary-path"
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  name="$(basename "$lib" .so)"; export NSR_HIP_LIB="$lib"
  for regime in "300 200" "5 20"; do
    set -- $regime; w=$1; st=$2
    python /root/repo/bench.py --steps $st --warmup $w $LEAN > "$out/${name}_w${w}.json" 2>/dev/null
    for attempt in 1 2 3; do
      rm -rf /tmp/pc && NSR_BENCH_NO_STEADY=1 NSR_BENCH_REGIME_OUT="$out/${name}_regime_w${w}.json" rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pc -o c -- python /root/repo/bench.py --steps $st --warmup $w $LEAN > /dev/null 2>&1
      f="$(find /tmp/pc -name '*counter_collection.csv' 2>/dev/null | head -1)"
      if [ -n "$f" ]; then python /root/repo/tools/pmc_summary.py "$f" FETCH_SIZE "$out/${name}_regime_w${w}.json" > "$out/${name}_fetch_w${w}.json" && break; fi
    done
  done
done
python - "$out" <<'PY'
import json, glob, os, sys
for f in sorted(glob.glob(sys.argv[1] + "/*_w*.json")):
    b = os.path.basename(f)
    try:
        if "_fetch_" in b:
            d = json.load(open(f)); print(b, {k[:28]: round(v["avg"] * 2 * 1024 / 1e6, 1) for k, v in d.items() if k.startswith(("k_grid_backward_owner", "k_mlp_wgrad"))})
        elif "_regime_" not in b:
            d = json.loads(open(f).read().strip().splitlines()[-1])
            print(b, round(d["ms_per_step"], 4), round(d["steady_state"]["ms_per_step"], 4), d["roofline"]["avg_launch_us"], d["kernels"]["hashgrid_backward_params"]["avg_us"])
    except Exception as e:
        print(b, "??", e)
PY
