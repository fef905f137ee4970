#!/bin/bash
# ON THE GPU BOX: end-to-end soaks of the NeuS trainers at the final code state (r03's schedule: 3,000 / 3,000 / 2,000 steps)
set -u
out=/root/repo/gpurun_out/r05v; mkdir -p "$out"
cd /root/repo
for c in "neus-blender 3000" "neus-dtu 3000" "neuralangelo 2000"; do
  set -- $c
  timeout 200 python tools/train_neus.py --config $1 --steps $2 2> "$out/soak_$1.err" | grep '^{' | tail -1 > "$out/soak_$1.json"
  cut -c1-400 "$out/soak_$1.json"
done
