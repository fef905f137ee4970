#!/bin/bash
# Baseline build for A/B runs on the GPU box: csrc/hashgrid.hip (+ its includes) as of a given commit, linked with the
# CURRENT objects of every other translation unit into build/variants/libnsr_hip_<tag>.so (pick it with NSR_HIP_LIB=<path>).
# Entry points the old source does not have are stubbed so that the ctypes binding table still resolves.
#   tools/build_baseline.sh <commit> <tag>
set -e
COMMIT="${1:?commit}"; TAG="${2:?tag}"
HERE="$(cd "$(dirname "$0")/.." && pwd)"
SRC="$(mktemp -d)/repo"; mkdir -p "$SRC/instant-nsr-pl_amd/csrc" "$SRC/include" "$HERE/build/variants"
for f in hashgrid.hip hashgrid_owner.inc hashgrid_geom.h nsr_common.h; do
  git -C "$HERE" show "$COMMIT:instant-nsr-pl_amd/csrc/$f" > "$SRC/instant-nsr-pl_amd/csrc/$f"
done
git -C "$HERE" show "$COMMIT:include/nsr_hip.h" > "$SRC/include/nsr_hip.h"
cat > "$SRC/stubs.cpp" <<'EOS'
#include <stdint.h>
extern "C" float nsr_hashgrid_owner_tune(int, float) { return -1.f; }
extern "C" int nsr_hashgrid_owner_debug_map(const void *, int, uint32_t, uint32_t, int, uint32_t *) { return -1; }
EOS
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wall -Wno-unused-function"
OUT="$HERE/build/variants"
hipcc $FLAGS -c "$SRC/instant-nsr-pl_amd/csrc/hashgrid.hip" -o "$OUT/hashgrid_$TAG.o"
g++ -O2 -fPIC -c "$SRC/stubs.cpp" -o "$OUT/stubs_$TAG.o"
CS="$HERE/instant-nsr-pl_amd/csrc"
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libnsr_hip_$TAG.so" "$OUT/hashgrid_$TAG.o" "$OUT/stubs_$TAG.o" \
  "$CS"/obj/{util,gridmlp,mlp,vmlp,neus,march,render,fused,occupancy,step}.o
echo "built $OUT/libnsr_hip_$TAG.so (hashgrid.hip @ $COMMIT)"
