#!/bin/bash
set -u
out=/root/repo/gpurun_out/r05f; mkdir -p "$out"
cd /root/repo
bash tools/fetch_calibration.sh "$out/fetch_calibration.json" 2>&1 | tail -3
timeout 900 python bench.py --no-other-workloads --no-boundary-path --no-cpu-baseline > "$out/bench_lean.json" 2> "$out/bench_lean.err"; tail -3 "$out/bench_lean.err"
python - "$out/bench_lean.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k in ("value", "ms_per_step", "host_enqueue_ms_per_step"):
    print(k, d[k])
print("regime", d["regime"]["kept_samples_per_step"], d["regime"]["marched_samples_per_step"])
print("steady", d["steady_state"] and {k: d["steady_state"][k] for k in ("ms_per_step", "samples_per_sec", "kept_samples_per_step_per_gpu")})
print("late", d["late_regime"])
print("whole", {k: v for k, v in (d["whole_run"] or {}).items() if k not in ("note", "test_views")})
print("roofline", {k: v for k, v in d["roofline"].items() if k not in ("traffic_source",)})
print("kernels", {k: round(v["avg_us"], 1) for k, v in d["kernels"].items()})
PY
