#!/bin/bash
set -u
out=/root/repo/gpurun_out/r05_final2; mkdir -p "$out"
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -q -x > "$out/pytest_gpu.log" 2>&1; tail -2 "$out/pytest_gpu.log"
python bench.py > "$out/bench_w20_s200.json" 2> "$out/bench_w20_s200.stderr"; tail -c 200 "$out/bench_w20_s200.json"; echo
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
cd /root/repo
python tools/forms_regime_ab.py 2>/dev/null | tail -1 > "$out/forms_regime_ab.json"
ls -la "$out" | head -20
