#!/bin/bash
cd /root/repo
LEAN="--no-cpu-baseline --no-other-workloads --no-boundary-path"
for m in 1 0 1; do
NSR_SIGMA_MODE=$m python bench.py --gpus 1 --steps 20 --warmup 5 $LEAN > gpurun_out/bench_sigma$m.json 2>gpurun_out/bench_sigma$m.err
python - $m <<'PY'
import json,sys
d=json.load(open('/root/repo/gpurun_out/bench_sigma%s.json'%sys.argv[1]))
print(sys.argv[1], {k:d[k] for k in ("value","ms_per_step","host_enqueue_ms_per_step")}, d["steady_state"]["ms_per_step"], d["transient"]["ms_per_step"], d["regime"]["kept_samples_per_step"], {k:(round(v["avg_launch_us"],1)) for k,v in d["kernels"].items()})
PY
done
