#!/bin/bash
# ON THE GPU BOX: runtime knobs around signals (what a cross-stream wait costs), separate processes, each under its own timeout
set -u
out=/root/repo/gpurun_out/r05u; mkdir -p "$out"
cd /root/repo
run() {
  tag=$1; shift
  env "$@" NSR_VARIANTS=round5_forms timeout 120 python tools/step_variants.py 700 160 4 > "$out/$tag.json" 2> "$out/$tag.err"
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
run sys_scope_signal_0 ROC_SYSTEM_SCOPE_SIGNAL=0
run default_2 NSR_DUMMY=1
run no_interrupt HSA_ENABLE_INTERRUPT=0
run sys_scope_signal_0_b ROC_SYSTEM_SCOPE_SIGNAL=0
run default_3 NSR_DUMMY=1
