#!/bin/bash
# Build libnsr_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [outdir]
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="${1:-$HERE/../nsr_hip}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wall -Wno-unused-function ${NSR_EXTRA_FLAGS}"
mkdir -p "$HERE/obj"
pids=()
for f in util hashgrid gridmlp mlp vmlp neus march render fused occupancy step; do
  ( hipcc $FLAGS -c "$HERE/$f.hip" -o "$HERE/obj/$f.o" ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libnsr_hip.so" "$HERE"/obj/{util,hashgrid,gridmlp,mlp,vmlp,neus,march,render,fused,occupancy,step}.o
echo "built $OUT/libnsr_hip.so"
