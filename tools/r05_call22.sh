#!/bin/bash
# ON THE GPU BOX: runtime knobs, separate processes, order A B A B: HIP_FORCE_DEV_KERNARG (kernel arguments in device memory)
set -u
out=/root/repo/gpurun_out/r05t; mkdir -p "$out"
cd /root/repo
run() {  # $1 tag, rest: env assignments
  tag=$1; shift
  env "$@" NSR_VARIANTS=round5_forms timeout 300 python tools/step_variants.py 700 160 4 > "$out/$tag.json" 2> "$out/$tag.err"
  python - "$out/$tag.json" "$tag" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    v = d["settings"]["round5_forms"]; print(f"{sys.argv[2]:28s} mean {v['mean_ms']:.4f} ms  {v['ms_per_step']}  host {v['host_ms_per_step']}")
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
run warmup NSR_DUMMY=1
run default_1 NSR_DUMMY=1
run dev_kernarg_1 HIP_FORCE_DEV_KERNARG=1
run default_2 NSR_DUMMY=1
run dev_kernarg_2 HIP_FORCE_DEV_KERNARG=1
run dev_kernarg_0 HIP_FORCE_DEV_KERNARG=0
