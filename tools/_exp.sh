#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r04_final2
timeout 300 python -m pytest tests/test_gpu_fused.py tests/test_capi.py -x -q 2>&1 | tail -2
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_final2/bench_w5_s20.json 2> gpurun_out/r04_final2/bench_w5_s20.stderr
timeout 400 python bench.py > gpurun_out/r04_final2/bench_w20_s200.json 2> gpurun_out/r04_final2/bench_w20_s200.stderr
python - <<'PY'
import json
for f in ("bench_w5_s20","bench_w20_s200"):
    d=json.load(open('/root/repo/gpurun_out/r04_final2/%s.json'%f))
    print(f, {k:d[k] for k in ("value","ms_per_step","host_enqueue_ms_per_step")}, d["steady_state"]["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], {k:round(d[k]["samples_per_sec"]/1e8,3) for k in ("boundary_path","boundary_path_neus","modular_path")}, {k:round(v["ms_per_step"],3) for k,v in d["other_workloads"].items()})
PY
