#!/bin/bash
set -u
out=/root/repo/gpurun_out/r05l; mkdir -p "$out"
cd /root/repo
for g in 2 1; do
  NSR_EXCHANGE_GROUPS=$g timeout 600 python tools/exchange_floor.py nerf-blender 2>/dev/null | grep "^{" | tail -1 > "$out/exchange_floor_groups$g.json"; cat "$out/exchange_floor_groups$g.json"
done
