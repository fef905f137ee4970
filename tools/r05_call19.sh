#!/bin/bash
set -u
out=/root/repo/gpurun_out/r05r; mkdir -p "$out"
cd /root/repo
for c in neus-blender neuralangelo; do
  python tools/neus_host_profile.py $c 100 > "$out/neus_host_profile_$c.txt" 2>&1
  grep -v Warning "$out/neus_host_profile_$c.txt" | head -34 | cut -c1-150
done
