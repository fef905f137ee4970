// Level geometry, corner indexing and the plain 8-corner trilinear encode of one (sample, level): the arithmetic every
// kernel that touches the hash grid shares (csrc/hashgrid.hip, csrc/gridmlp.hip).  tcnn's published grid_index / hash
// (reference call sites models/network_utils.py:47,90,209).
#pragma once
#include "nsr_common.h"

namespace {

constexpr uint32_t PRIME_Y = 2654435761u;
constexpr uint32_t PRIME_Z = 805459861u;

struct LevelGeom {
    float scale;
    uint32_t res;
    uint32_t size;
    uint32_t offset;
    bool dense;
};

__device__ __forceinline__ LevelGeom load_level(const NsrGridDesc &d, uint32_t level)
{
    LevelGeom g;
    g.scale = d.scale[level];
    g.res = d.resolution[level];
    g.size = d.size[level];
    g.offset = d.offset[level];
    g.dense = (uint64_t)g.res * g.res * g.res <= (uint64_t)g.size;
    return g;
}

// entry index of an integer corner (tcnn grid_index: x-fastest dense while the stride fits, else hash)
__device__ __forceinline__ uint32_t corner_index(const LevelGeom &g, uint32_t cx, uint32_t cy, uint32_t cz)
{
    if (g.dense) {
        uint32_t idx = cx + g.res * (cy + g.res * cz);
        return idx >= g.size ? idx % g.size : idx;  // only the x==1.0 border can exceed
    }
    uint32_t h = cx ^ (cy * PRIME_Y) ^ (cz * PRIME_Z);
    return h & (g.size - 1);  // hashed levels are capped at T = 2^log2 entries
}

struct Cell {
    float w[3];      // fractional position inside the cell
    uint32_t c[3];   // integer corner (lower)
};

__device__ __forceinline__ Cell locate(const LevelGeom &g, float x0, float x1, float x2)
{
    Cell c;
    const float p0 = fmaf(g.scale, x0, 0.5f), p1 = fmaf(g.scale, x1, 0.5f), p2 = fmaf(g.scale, x2, 0.5f);
    const float f0 = floorf(p0), f1 = floorf(p1), f2 = floorf(p2);
    c.w[0] = p0 - f0; c.w[1] = p1 - f1; c.w[2] = p2 - f2;
    c.c[0] = (uint32_t)(int)f0; c.c[1] = (uint32_t)(int)f1; c.c[2] = (uint32_t)(int)f2;
    return c;
}

template <int F> struct FeatVec;
template <> struct FeatVec<1> { using T = __half; };
template <> struct FeatVec<2> { using T = __half2; };

template <int F>
__device__ __forceinline__ void load_feat(const __half *__restrict__ table, uint32_t entry, float (&v)[F])
{
    const __half *p = table + (uint64_t)entry * F;
    if constexpr (F == 1) {
        v[0] = __half2float(p[0]);
    } else if constexpr (F == 2) {
        const __half2 h = *reinterpret_cast<const __half2 *>(p);
        v[0] = __low2float(h); v[1] = __high2float(h);
    } else if constexpr (F == 4) {
        const uint2 raw = *reinterpret_cast<const uint2 *>(p);
        const __half2 a = *reinterpret_cast<const __half2 *>(&raw.x), b = *reinterpret_cast<const __half2 *>(&raw.y);
        v[0] = __low2float(a); v[1] = __high2float(a); v[2] = __low2float(b); v[3] = __high2float(b);
    } else {
        const uint4 raw = *reinterpret_cast<const uint4 *>(p);
        const uint32_t r[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const __half2 a = *reinterpret_cast<const __half2 *>(&r[k]);
            v[2 * k] = __low2float(a); v[2 * k + 1] = __high2float(a);
        }
    }
}

template <int F>
__device__ __forceinline__ void encode_level_from(const __half *__restrict__ tbl /* level base, global or LDS */,
                                                  const LevelGeom &g, float x0, float x1, float x2, float (&acc)[F])
{
    const Cell c = locate(g, x0, x1, x2);
#pragma unroll
    for (int f = 0; f < F; ++f) acc[f] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t e = corner_index(g, c.c[0] + (k & 1), c.c[1] + ((k >> 1) & 1), c.c[2] + ((k >> 2) & 1));
        float v[F];
        load_feat<F>(tbl, e, v);
        float w = (k & 1) ? c.w[0] : 1.f - c.w[0];
        w *= (k & 2) ? c.w[1] : 1.f - c.w[1];
        w *= (k & 4) ? c.w[2] : 1.f - c.w[2];
#pragma unroll
        for (int f = 0; f < F; ++f) acc[f] = fmaf(w, v[f], acc[f]);
    }
}

}  // namespace
