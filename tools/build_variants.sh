#!/bin/bash
# A/B builds of the owner-computes table backward: csrc/hashgrid.hip compiled with different tuning macros and linked with
# the other (unchanged) objects into build/variants/libnsr_hip_<name>.so ; pick one at run time with NSR_HIP_LIB=<path>.
#   tools/build_variants.sh name1:"-DNSR_OWN_BATCH=4" name2:"-DNSR_OWN_LOG2=12 -DNSR_OWN_BLOCK=512" ...
set -e
HERE="$(cd "$(dirname "$0")/.." && pwd)"
CS="$HERE/instant-nsr-pl_amd/csrc"
OUT="$HERE/build/variants"; mkdir -p "$OUT"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wall -Wno-unused-function"
pids=()
for spec in "$@"; do
  name="${spec%%:*}"; defs="${spec#*:}"
  ( hipcc $FLAGS $defs -c "$CS/hashgrid.hip" -o "$OUT/hashgrid_$name.o" &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libnsr_hip_$name.so" "$OUT/hashgrid_$name.o" \
      "$CS"/obj/{util,gridmlp,mlp,vmlp,neus,march,render,fused,occupancy,step}.o && echo "built $name ($defs)" ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
