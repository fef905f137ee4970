#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the round-6 evidence set.  bench lines for both command lines (default = --warmup 20
# --steps 200 behind 300 untimed set-up steps; the driver's --warmup 5 --steps 20), rocprofv3 kernel summaries + timeline for
# both, PMC FETCH / WRITE passes for both (the FETCH calibration sweep first), the secondary kernels' PMC passes, kernel
# breakdowns of the three NeuS workloads at the reference's operating point, the fp32 MLP / small-kernel micro-benchmarks,
# the step's forms A/B.  Writes small summaries under gpurun_out/$1/ ; copy what should be judged into profiles/
# (tools/install_profiles_r06.sh).
set -u
tag="${1:-r06_final}"; out="/root/repo/gpurun_out/$tag"; mkdir -p "$out"
cd /root/repo
bash tools/fetch_calibration.sh "$out/fetch_calibration.json" > /dev/null 2>&1
python bench.py > "$out/bench_w20_s200.json" 2> "$out/bench_w20_s200.stderr"; tail -c 300 "$out/bench_w20_s200.json"; echo
python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_w5_s20.json" 2> "$out/bench_w5_s20.stderr"; tail -c 200 "$out/bench_w5_s20.json"; echo
LEAN="--no-cpu-baseline --no-other-workloads --no-boundary-path --no-whole-run"
export NSR_BENCH_NO_STEADY=1 NSR_BENCH_NO_FORMS_AB=1
cd /tmp && export TMPDIR=/tmp
for regime in "20 200" "5 20"; do
  set -- $regime; w=$1; st=$2
  rm -rf /tmp/pk && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o k -- python /root/repo/bench.py --steps $st --warmup $w $LEAN > "$out/bench_under_rocprof_w${w}_s${st}.json" 2>/dev/null
  cp "$(find /tmp/pk -name '*kernel_stats.csv' | head -1)" "$out/kernel_stats_w${w}_s${st}.csv"
  python /root/repo/tools/trace_tail.py "$(find /tmp/pk -name '*kernel_trace.csv' | head -1)" "$out/timeline_tail_w${w}_s${st}.csv" 7000
done
for regime in "20 200" "5 20"; do
  set -- $regime; w=$1; st=$2; rd="$out/pmc_w${w}_s${st}"; mkdir -p "$rd"
  for c in FETCH_SIZE WRITE_SIZE; do
    for attempt in 1 2 3 4; do  # rocprofv3 --pmc occasionally segfaults at exit on this image: retry, the passes are independent
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
unset NSR_BENCH_NO_STEADY NSR_BENCH_NO_FORMS_AB
bash /root/repo/tools/secondary_pmc.sh "$tag" > /dev/null 2>&1
for c in neus-blender neus-dtu neuralangelo; do
  python /root/repo/tools/neus_operating_point.py $c 100 2>/dev/null | tail -1 > "$out/neus_op_$c.json"
  rm -rf /tmp/pn && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pn -o k -- python /root/repo/tools/neus_operating_point.py $c 60 > /dev/null 2>&1
  cp "$(find /tmp/pn -name '*kernel_stats.csv' | head -1)" "$out/neus_op_${c}_kernel_stats.csv"
done
# MFMA-busy share of the fp32 MLP kernels (C5 shapes): separate --pmc passes, kernel trace only
for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES; do
  for attempt in 1 2 3; do
    rm -rf /tmp/pv && NO_CHECK=1 timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pv -o v -- python /root/repo/tools/vmlp_bench.py > /dev/null 2>&1
    f="$(find /tmp/pv -name '*counter_collection.csv' 2>/dev/null | head -1)"
    if [ -n "$f" ]; then python /root/repo/tools/pmc_summary.py "$f" $c > "$out/vmlp_pmc_$c.json" && break; fi
  done
done
cd /root/repo
python tools/vmlp_bench.py > "$out/vmlp_bench.json" 2>/dev/null
python tools/small_kernels_bench.py > "$out/small_kernels.json" 2>/dev/null
python tools/kernel_microbench.py > "$out/microbench.json" 2>/dev/null
python tools/step_variants.py 2500 160 4 > "$out/step_variants_2500.json" 2>/dev/null
python tools/step_variants.py 450 160 4 > "$out/step_variants_450.json" 2>/dev/null
python tools/late_regime.py 10000 1000 > "$out/late_regime.json" 2>/dev/null
ls -la "$out"
