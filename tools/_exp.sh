#!/bin/bash
cd /root/repo
LEAN="--no-cpu-baseline --no-other-workloads --no-boundary-path"
for i in 1 2 3; do
python bench.py --gpus 1 --steps 20 --warmup 5 $LEAN > gpurun_out/bench_cold$i.json 2>gpurun_out/bench_cold$i.err
python - $i <<'PY'
import json,sys
d=json.load(open('/root/repo/gpurun_out/bench_cold%s.json'%sys.argv[1]))
print(sys.argv[1], {k:d[k] for k in ("value","ms_per_step","host_enqueue_ms_per_step")}, d["steady_state"]["ms_per_step"], d["transient"]["ms_per_step"], d["regime"]["kept_samples_per_step"])
PY
done
