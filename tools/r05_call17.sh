#!/bin/bash
# ON THE GPU BOX: the ray-ordered sigma pass late in training (82 % of the marched samples behind their ray's cut), same process A/B
set -u
out=/root/repo/gpurun_out/r05q; mkdir -p "$out"
cd /root/repo
for at in 10000 2500; do
  NSR_VARIANTS=round5_forms,round5_sigma_rays timeout 600 python tools/step_variants.py $at 160 4 > "$out/sigma_rays_$at.json" 2> "$out/sigma_rays_$at.err"
  python - "$out/sigma_rays_$at.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d["settings"].items():
    print(f"{k:24s} mean {v['mean_ms']:.4f} ms  {v['ms_per_step']}  kept {v['kept_per_step']}")
PY
done
