// Multiresolution hash-grid encoding for gfx950 (replaces tcnn.Encoding(HashGrid); reference call sites
// models/network_utils.py:47,90,209 ; models/geometry.py:124,169,177-180,195).
//
// Work decomposition (MI355X-first, not tcnn's (N/512, L) grid):
//   * one lane = one (sample, level); a wavefront = 64 consecutive samples of ONE level, so the x-loads
//     are coalesced and the 8 corner gathers of neighbouring samples (ray-ordered => spatially coherent)
//     fall into the same or adjacent 64-B sectors of that level's table;
//   * the 1-D grid is XCD-aware: hardware places block b on XCD (b % 8) and every XCD has a private
//     4 MiB L2, so level l is only ever touched by XCD (l % 8).  With L=16, T=2^19, F=2 (fp16) an XCD
//     serves two levels = 4 MiB of table: the fine levels stay L2-resident instead of thrashing all
//     eight L2s with the whole 24 MiB table.  Placement is a speed assumption only, never correctness.
//   * level geometry (scale/resolution/size/offset) arrives precomputed in fp32 from the host
//     (NsrGridDesc) and is read through scalar loads (block-uniform level).
//   * table gradients: NO global float atomics on the default path.  (sample, corner-pair) items are binned by the
//     owning slice of a level's gradient; a workgroup accumulates only its own items into a 128 KiB LDS slice in Q27.36
//     fixed point (64-bit integer LDS atomics) and stores the slice once as fp32 ("owner computes", see below).  The
//     one-lane-per-(sample, level) kernel with global_atomic_add_f32 is kept for the API without a workspace.
//   * the forward optionally keeps the per-level Jacobian d y / d x (k_grid_forward with `jac`) so that the analytic
//     NeuS normal and its double backward are dense products instead of a second and a third table gather, and
//     k_grid_forward_taps encodes a sample together with its six finite-difference taps from shared corner loads.
#include <string.h>

#include "nsr_common.h"

namespace {

constexpr uint32_t PRIME_Y = 2654435761u;
constexpr uint32_t PRIME_Z = 805459861u;
constexpr int GRID_BLOCK = 256;

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

template <bool F32> struct GradT { using T = __half; };
template <> struct GradT<true> { using T = float; };

template <bool F32>
__device__ __forceinline__ float load_grad(const void *p, uint64_t i)
{
    if constexpr (F32) return reinterpret_cast<const float *>(p)[i];
    else return __half2float(reinterpret_cast<const __half *>(p)[i]);
}

// XCD-aware (block -> level, sample-block) mapping.  lpx = ceil(L/8) levels per XCD.
__device__ __forceinline__ bool map_block(uint32_t n_levels, uint32_t lpx, uint32_t &level, uint32_t &blk)
{
    const uint32_t b = blockIdx.x, xcd = b & 7u, q = b >> 3;
    level = xcd + 8u * (q % lpx);
    blk = q / lpx;
    return level < n_levels;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int F>
__global__ void __launch_bounds__(GRID_BLOCK)
k_grid_forward(const float *__restrict__ x, const __half *__restrict__ table, __half *__restrict__ y, uint32_t n,
               uint32_t y_stride, uint32_t mask_count, uint32_t lpx, int level_major, const NsrGridDesc d,
               const int32_t *__restrict__ n_dev, float *__restrict__ jac /* [L][n][F][3] d y / d x, or NULL */,
               uint32_t level_begin /* levels below it are produced by k_grid_forward_lds */)
{
    uint32_t level, blk;
    if (!map_block(d.n_levels, lpx, level, blk)) return;
    if (level < level_begin) return;
    const uint32_t i = blk * GRID_BLOCK + threadIdx.x;
    if (i >= live_count(n, n_dev)) return;
    // row-major [n, y_stride] is what the tcnn API returns; level-major [L][n][F] is what the fused path uses: a wave
    // then stores 64 x F consecutive halfs (measured: the row-major 4-B stores, issued level by level from different
    // XCDs, cost 187 MB of fabric writes for a 19 MB output)
    __half *yo = level_major ? y + ((uint64_t)level * n + i) * F : y + (uint64_t)i * y_stride + level * F;
    float acc[F];
#pragma unroll
    for (int f = 0; f < F; ++f) acc[f] = 0.f;
    if (level < mask_count) {
        const LevelGeom g = load_level(d, level);
        const Cell c = locate(g, x[3ull * i], x[3ull * i + 1], x[3ull * i + 2]);
        float v[8][F];
        bool paired = false;
        if constexpr (F == 2) {
            // Hashed level, even cell x: the x-neighbour's slot is hash(x | 1, y, z) = hash(x, y, z) ^ 1 -- the other half of
            // the same aligned 8-byte pair.  ONE 8-B gather then serves both corners.  The forward is bound by the L2 request
            // rate (88 scattered 4-B gathers per sample over the 11 hashed levels); this removes a quarter of them.
            paired = !g.dense && !(c.c[0] & 1u);
            if (paired) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t e0 = corner_index(g, c.c[0], c.c[1] + (j & 1), c.c[2] + (j >> 1));
                    const uint2 raw = *reinterpret_cast<const uint2 *>(table + (uint64_t)(g.offset + (e0 & ~1u)) * 2);
                    const __half2 lo = *reinterpret_cast<const __half2 *>(&raw.x), hi = *reinterpret_cast<const __half2 *>(&raw.y);
                    const __half2 a = (e0 & 1u) ? hi : lo, b = (e0 & 1u) ? lo : hi;  // entry e0, entry e0 ^ 1
                    v[2 * j][0] = __low2float(a); v[2 * j][1] = __high2float(a);
                    v[2 * j + 1][0] = __low2float(b); v[2 * j + 1][1] = __high2float(b);
                }
            }
        }
        if (!paired) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t e = corner_index(g, c.c[0] + (k & 1), c.c[1] + ((k >> 1) & 1), c.c[2] + ((k >> 2) & 1));
                load_feat<F>(table, g.offset + e, v[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float w = (k & 1) ? c.w[0] : 1.f - c.w[0];
            w *= (k & 2) ? c.w[1] : 1.f - c.w[1];
            w *= (k & 4) ? c.w[2] : 1.f - c.w[2];
#pragma unroll
            for (int f = 0; f < F; ++f) acc[f] = fmaf(w, v[k][f], acc[f]);
        }
        if (jac) {
            // The Jacobian of this level's F outputs w.r.t. x, from the 8 corner values already in registers: the input
            // gradient (J^T dy) and its double backward (J g) then need no second / third gather of the table
            // (what tcnn's dy_dx buffer is for; models/geometry.py:176-180 asks for both every NeuS step)
            float *jo = jac + ((uint64_t)level * n + i) * (F * 3);
            const float w0 = c.w[0], w1 = c.w[1], w2 = c.w[2];
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const float d0 = (1.f - w1) * (1.f - w2) * (v[1][f] - v[0][f]) + w1 * (1.f - w2) * (v[3][f] - v[2][f]) +
                                 (1.f - w1) * w2 * (v[5][f] - v[4][f]) + w1 * w2 * (v[7][f] - v[6][f]);
                const float d1 = (1.f - w0) * (1.f - w2) * (v[2][f] - v[0][f]) + w0 * (1.f - w2) * (v[3][f] - v[1][f]) +
                                 (1.f - w0) * w2 * (v[6][f] - v[4][f]) + w0 * w2 * (v[7][f] - v[5][f]);
                const float d2 = (1.f - w0) * (1.f - w1) * (v[4][f] - v[0][f]) + w0 * (1.f - w1) * (v[5][f] - v[1][f]) +
                                 (1.f - w0) * w1 * (v[6][f] - v[2][f]) + w0 * w1 * (v[7][f] - v[3][f]);
                jo[f * 3] = g.scale * d0; jo[f * 3 + 1] = g.scale * d1; jo[f * 3 + 2] = g.scale * d2;
            }
        }
    } else if (jac) {
        float *jo = jac + ((uint64_t)level * n + i) * (F * 3);
#pragma unroll
        for (int q = 0; q < F * 3; ++q) jo[q] = 0.f;
    }
    if constexpr (F == 1) {
        yo[0] = __float2half_rn(acc[0]);
    } else {
#pragma unroll
        for (int f = 0; f < F; f += 2)
            *reinterpret_cast<__half2 *>(yo + f) = __floats2half2_rn(acc[f], acc[f + 1]);
    }
}

// ------------------------------------------------------------------------------------------------
// Forward variants for the A/B the north-star asks for (profiles/r02_forward_ab.json):
//   * k_grid_forward_lds : the table of a small dense level (<= 16384 entries: 64 KB at F = 2) is staged in LDS by a
//     persistent workgroup that then encodes a contiguous chunk of samples for that level from LDS;
//   * k_grid_forward_pair: one lane encodes BOTH levels its XCD owns (l and l + 8): x is loaded once, 16 gathers in flight.
// Selected at run time by nsr_hashgrid_forward_variant(); the default is (0, 2) for grids of <= 16 levels.
// ------------------------------------------------------------------------------------------------
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

template <int F>
__device__ __forceinline__ void store_enc(__half *yo, const float (&acc)[F])
{
    if constexpr (F == 1) {
        yo[0] = __float2half_rn(acc[0]);
    } else {
#pragma unroll
        for (int f = 0; f < F; f += 2) *reinterpret_cast<__half2 *>(yo + f) = __floats2half2_rn(acc[f], acc[f + 1]);
    }
}

constexpr int LDS_FWD_BLOCK = 1024;
template <int F>
__global__ void __launch_bounds__(LDS_FWD_BLOCK)
k_grid_forward_lds(const float *__restrict__ x, const __half *__restrict__ table, __half *__restrict__ y, uint32_t n,
                   uint32_t y_stride, uint32_t mask_count, int level_major, uint32_t blocks_per_level,
                   const NsrGridDesc d, const int32_t *__restrict__ n_dev)
{
    extern __shared__ __attribute__((aligned(16))) __half lds_table[];
    const uint32_t level = blockIdx.x / blocks_per_level, part = blockIdx.x % blocks_per_level;
    const uint32_t n_live = live_count(n, n_dev);
    const LevelGeom g = load_level(d, level);
    if (level < mask_count) {
        const uint4 *src = reinterpret_cast<const uint4 *>(table + (uint64_t)g.offset * F);
        uint4 *dst = reinterpret_cast<uint4 *>(lds_table);
        const uint32_t n16 = (g.size * F * 2 + 15) / 16;  // level sizes are multiples of 8 entries
        for (uint32_t k = threadIdx.x; k < n16; k += LDS_FWD_BLOCK) dst[k] = src[k];
    }
    __syncthreads();
    const uint32_t per = (n_live + blocks_per_level - 1) / blocks_per_level;
    const uint32_t i0 = part * per, i1 = min(n_live, i0 + per);
    for (uint32_t i = i0 + threadIdx.x; i < i1; i += LDS_FWD_BLOCK) {
        float acc[F];
        if (level < mask_count) {
            encode_level_from<F>(lds_table, g, x[3ull * i], x[3ull * i + 1], x[3ull * i + 2], acc);
        } else {
#pragma unroll
            for (int f = 0; f < F; ++f) acc[f] = 0.f;
        }
        store_enc<F>(level_major ? y + ((uint64_t)level * n + i) * F : y + (uint64_t)i * y_stride + level * F, acc);
    }
}

// blocks b with (b & 7) = xcd, q = b >> 3: both levels {xcd, xcd + 8} of sample block q (levels < level_begin skipped)
template <int F>
__global__ void __launch_bounds__(GRID_BLOCK)
k_grid_forward_pair(const float *__restrict__ x, const __half *__restrict__ table, __half *__restrict__ y, uint32_t n,
                    uint32_t y_stride, uint32_t mask_count, uint32_t level_begin, int level_major, const NsrGridDesc d,
                    const int32_t *__restrict__ n_dev)
{
    const uint32_t xcd = blockIdx.x & 7u, blk = blockIdx.x >> 3;
    const uint32_t i = blk * GRID_BLOCK + threadIdx.x;
    if (i >= live_count(n, n_dev)) return;
    const float x0 = x[3ull * i], x1 = x[3ull * i + 1], x2 = x[3ull * i + 2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint32_t level = xcd + 8u * h;
        if (level >= d.n_levels || level < level_begin) continue;
        float acc[F];
        if (level < mask_count) {
            const LevelGeom g = load_level(d, level);
            const Cell c = locate(g, x0, x1, x2);
            const __half *tbl = table + (uint64_t)g.offset * F;
            float v[8][F];
            bool paired = false;
            if constexpr (F == 2) {  // same paired 8-B gathers as k_grid_forward
                paired = !g.dense && !(c.c[0] & 1u);
                if (paired) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t e0 = corner_index(g, c.c[0], c.c[1] + (j & 1), c.c[2] + (j >> 1));
                        const uint2 raw = *reinterpret_cast<const uint2 *>(tbl + (uint64_t)(e0 & ~1u) * 2);
                        const __half2 lo = *reinterpret_cast<const __half2 *>(&raw.x), hi = *reinterpret_cast<const __half2 *>(&raw.y);
                        const __half2 a = (e0 & 1u) ? hi : lo, b = (e0 & 1u) ? lo : hi;
                        v[2 * j][0] = __low2float(a); v[2 * j][1] = __high2float(a);
                        v[2 * j + 1][0] = __low2float(b); v[2 * j + 1][1] = __high2float(b);
                    }
                }
            }
            if (!paired) {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    load_feat<F>(tbl, corner_index(g, c.c[0] + (k & 1), c.c[1] + ((k >> 1) & 1), c.c[2] + ((k >> 2) & 1)), v[k]);
            }
#pragma unroll
            for (int f = 0; f < F; ++f) acc[f] = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float w = (k & 1) ? c.w[0] : 1.f - c.w[0];
                w *= (k & 2) ? c.w[1] : 1.f - c.w[1];
                w *= (k & 4) ? c.w[2] : 1.f - c.w[2];
#pragma unroll
                for (int f = 0; f < F; ++f) acc[f] = fmaf(w, v[k][f], acc[f]);
            }
        } else {
#pragma unroll
            for (int f = 0; f < F; ++f) acc[f] = 0.f;
        }
        store_enc<F>(level_major ? y + ((uint64_t)level * n + i) * F : y + (uint64_t)i * y_stride + level * F, acc);
    }
}

// ------------------------------------------------------------------------------------------------
// forward of a sample AND its six finite-difference taps (reference models/geometry.py:181-197: the neuralangelo
// configs evaluate the encoder at x and x +- eps e_k, eps = one cell of the finest ACTIVE level).  x7: [7][n][3] unit
// coordinates, rows 1 + 2k / 2 + 2k = the +eps / -eps tap along axis k; only that axis differs from the sample, and on
// every active level (cell >= eps) a tap sits in the sample's cell or in the neighbour across ONE face.  So a lane
// gathers the sample's 8 corners once and per tap only the 4 corners of the far face when the tap crossed it
// (expected 8 + ~6 gathers per level over 16 active levels instead of 56): one position load, 7 encodes.
// ------------------------------------------------------------------------------------------------
template <int F>
__device__ __forceinline__ void blend8(const Cell &c, const float (&v)[8][F], __half *yo)
{
    float acc[F];
#pragma unroll
    for (int f = 0; f < F; ++f) acc[f] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float w = (k & 1) ? c.w[0] : 1.f - c.w[0];
        w *= (k & 2) ? c.w[1] : 1.f - c.w[1];
        w *= (k & 4) ? c.w[2] : 1.f - c.w[2];
#pragma unroll
        for (int f = 0; f < F; ++f) acc[f] = fmaf(w, v[k][f], acc[f]);
    }
    if constexpr (F == 1) {
        yo[0] = __float2half_rn(acc[0]);
    } else {
#pragma unroll
        for (int f = 0; f < F; f += 2) *reinterpret_cast<__half2 *>(yo + f) = __floats2half2_rn(acc[f], acc[f + 1]);
    }
}

template <int F>
__global__ void __launch_bounds__(GRID_BLOCK)
k_grid_forward_taps(const float *__restrict__ x7, const __half *__restrict__ table, __half *__restrict__ y, uint32_t n,
                    uint32_t y_stride, uint32_t mask_count, uint32_t lpx, int level_major, const NsrGridDesc d,
                    const int32_t *__restrict__ n_dev)
{
    // row pointer of point p (0 .. 7n-1): row-major [7n][y_stride] or level-major [L][7n][F]
    const uint64_t n7 = 7ull * n;
#define TAP_ROW(p) (level_major ? y + ((uint64_t)level * n7 + (p)) * F : y + (uint64_t)(p) * y_stride + level * F)
    uint32_t level, blk;
    if (!map_block(d.n_levels, lpx, level, blk)) return;
    const uint32_t i = blk * GRID_BLOCK + threadIdx.x;
    if (i >= live_count(n, n_dev)) return;
    if (level >= mask_count) {
#pragma unroll
        for (int t = 0; t < 7; ++t)
#pragma unroll
            for (int f = 0; f < F; ++f) TAP_ROW((uint64_t)t * n + i)[f] = __float2half_rn(0.f);
        return;
    }
    const LevelGeom g = load_level(d, level);
    const float xb[3] = {x7[3ull * i], x7[3ull * i + 1], x7[3ull * i + 2]};
    const Cell cb = locate(g, xb[0], xb[1], xb[2]);
    float vb[8][F];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        load_feat<F>(table, g.offset + corner_index(g, cb.c[0] + (k & 1), cb.c[1] + ((k >> 1) & 1), cb.c[2] + ((k >> 2) & 1)),
                     vb[k]);
    blend8<F>(cb, vb, TAP_ROW(i));
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const int a = t >> 1;  // the axis this tap moved along
        float xt[3] = {xb[0], xb[1], xb[2]};
        xt[a] = x7[((uint64_t)(t + 1) * n + i) * 3 + a];
        const Cell ct = locate(g, xt[0], xt[1], xt[2]);
        const int dc = (int)ct.c[a] - (int)cb.c[a];
        float vt[8][F];
        if (dc == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int f = 0; f < F; ++f) vt[k][f] = vb[k][f];
        } else if (dc == 1 || dc == -1) {
            // corner k of the tap's cell with bit a == (dc < 0) is corner k ^ (1 << a) of the sample's cell (shared face)
            const int shared_bit = dc < 0 ? 1 : 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (((k >> a) & 1) == shared_bit) {
#pragma unroll
                    for (int f = 0; f < F; ++f) vt[k][f] = vb[k ^ (1 << a)][f];
                } else {
                    load_feat<F>(table, g.offset + corner_index(g, ct.c[0] + (k & 1), ct.c[1] + ((k >> 1) & 1),
                                                                ct.c[2] + ((k >> 2) & 1)), vt[k]);
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                load_feat<F>(table, g.offset + corner_index(g, ct.c[0] + (k & 1), ct.c[1] + ((k >> 1) & 1),
                                                            ct.c[2] + ((k >> 2) & 1)), vt[k]);
        }
        blend8<F>(ct, vt, TAP_ROW((uint64_t)(t + 1) * n + i));
    }
#undef TAP_ROW
}

// ------------------------------------------------------------------------------------------------
// backward w.r.t. the table: scatter-add with fp32 hardware atomics
// ------------------------------------------------------------------------------------------------
template <int F, bool DY_F32>
__global__ void __launch_bounds__(GRID_BLOCK)
k_grid_backward_params(const float *__restrict__ x, const void *__restrict__ dy, uint32_t dy_stride,
                       float *__restrict__ grad_table, uint32_t n, uint32_t mask_count, uint32_t lpx,
                       float grad_scale, const NsrGridDesc d)
{
    uint32_t level, blk;
    if (!map_block(d.n_levels, lpx, level, blk)) return;
    if (level >= mask_count) return;
    const uint32_t i = blk * GRID_BLOCK + threadIdx.x;
    if (i >= n) return;
    float g_out[F];
    bool any = false;
#pragma unroll
    for (int f = 0; f < F; ++f) {
        g_out[f] = load_grad<DY_F32>(dy, (uint64_t)i * dy_stride + level * F + f) * grad_scale;
        any |= (g_out[f] != 0.f);
    }
    if (!any) return;
    const LevelGeom g = load_level(d, level);
    const Cell c = locate(g, x[3ull * i], x[3ull * i + 1], x[3ull * i + 2]);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t e = corner_index(g, c.c[0] + (k & 1), c.c[1] + ((k >> 1) & 1), c.c[2] + ((k >> 2) & 1));
        float w = (k & 1) ? c.w[0] : 1.f - c.w[0];
        w *= (k & 2) ? c.w[1] : 1.f - c.w[1];
        w *= (k & 4) ? c.w[2] : 1.f - c.w[2];
        float *gp = grad_table + (uint64_t)(g.offset + e) * F;
#pragma unroll
        // (measured: workgroup-scope atomics compile to the same global_atomic_add_f32 and run at the same 22 G/s --
        //  the memory mapping, not the scope bits, sends them to the memory side; there is no "L2-local" shortcut)
        for (int f = 0; f < F; ++f) unsafeAtomicAdd(gp + f, w * g_out[f]);
    }
}


// ------------------------------------------------------------------------------------------------
// backward w.r.t. the table, "owner computes" (the default): NO global atomics.
//
// Measured on MI355X: device-scope fp32 atomics retire at ~22 G/s chip-wide (they execute memory-side so
// that the 8 non-coherent XCD L2s stay consistent) -- 1.13 ms for one 98 k-sample step, 3x everything else.
// Rays of a batch are i.i.d., so fine-level entries are touched ~once per step: there is no reuse to cache,
// only the atomic *rate* hurts.  So instead every workgroup OWNS a contiguous slice of one level's gradient
// that fits the CU's 160 KiB LDS (10240 64-bit pairs), scans ALL samples of that level, recomputes the 8
// corner indices (a few dozen VALU ops -- the chip has ~100x more VALU than atomic throughput), accumulates
// the corners that fall into its slice with 64-bit fixed-point LDS atomics (ds_add_u64) and finally stores the slice
// as fp32 with plain coalesced 16-B stores.  ~750 workgroups cover L=16,T=2^19,F=2; each grad entry is written exactly once, so the
// 50 MB gradient needs no memset either (accumulate=0).  dy is read level-major ([L][N][F], 8-B coalesced).
// ------------------------------------------------------------------------------------------------
// slice size / workgroup size, measured at the NeRF step's operating point (tools/table_backward_variants.py, 9.6e4 coherent
// samples, accumulate / accumulate + AdamW): 2^13 entries x 1024 threads (one workgroup per CU) 87 / 122 us, 2^12 x 512
// 77 / 111 us, 2^11 x 256 72 / 105 us -- several small workgroups per CU overlap one's item phase (LDS atomics, latency
// bound) with another's write-out / AdamW phase (HBM streaming).  More items in flight per lane (batch 4, 8) changed nothing.
#ifndef NSR_OWN_BLOCK
#define NSR_OWN_BLOCK 256
#endif
#ifndef NSR_OWN_LOG2
#define NSR_OWN_LOG2 11
#endif
#ifndef NSR_OWN_BIN_SPT
#define NSR_OWN_BIN_SPT 4
#endif
constexpr int OWN_BLOCK = NSR_OWN_BLOCK;
constexpr int OWN_POW2_LOG2 = NSR_OWN_LOG2;            // hashed levels: 8192-entry slices (128 KiB at F=2) -> owner = hash bits
constexpr int OWN_LDS_WORDS = 2 << NSR_OWN_LOG2;       // 64-bit accumulators (measured: 2^13 119 us, 2^12 133 us, 2^11 124 us)
constexpr int OWN_TARGET_WGS = 32;      // workgroups per level the decomposition aims for
#ifndef NSR_OWN_RL_MAX
#define NSR_OWN_RL_MAX 10
#endif
#ifndef NSR_OWN_DENSE_WGS
#define NSR_OWN_DENSE_WGS 64
#endif
#ifndef NSR_OWN_BATCH
#define NSR_OWN_BATCH 2
#endif
constexpr int OWN_DENSE_TARGET_WGS = NSR_OWN_DENSE_WGS; // ... for the dense (coarse) levels: every sample lands in few entries and the
                                         // workgroups serialise on LDS conflicts -- more, smaller item chunks
                                         // (measured at 1.28e5 surface samples: 32 -> 185 us, 64 -> 177 us, 128 -> 198 us)
constexpr int OWN_MAX_SLICES = 2048;    // per level (T = 2^24 at 8192-entry slices)
constexpr int OWN_BIN_BLOCK = 256;      // threads per block of the two binning passes
constexpr int OWN_BIN_SPT = NSR_OWN_BIN_SPT;  // samples per thread: fewer, larger blocks -> fewer global range reservations
constexpr float OWN_FIX_SCALE = 68719476736.f;          // 2^36: accumulators are Q27.36 fixed point
constexpr float OWN_FIX_INV = 1.f / 68719476736.f;

// Decomposition of one level: R slices of its gradient x C item chunks.  C > 1 (small dense levels, where one slice
// would take every sample and serialise on same-address LDS atomics) writes per-chunk slabs that a second tiny kernel
// sums; C == 1 stores straight into the gradient.
struct OwnerMap {
    // block b runs on XCD b % 8 (round-robin dispatch).  The workgroups of every level are dealt to ALL XCDs: keeping a
    // level on one XCD (its L2 then serves dy to every slice) was measured slower -- 11 hashed levels do not divide
    // over 8 XCDs, and the slowest XCD sets the kernel time (1.28e5 samples: 201 us whole-level vs 177 us dealt).
    uint32_t level_start[NSR_MAX_LEVELS];  // first row (b / 8) of the level; rows [start, start + ceil(wgs / 8))
    uint32_t n_slices[NSR_MAX_LEVELS];
    uint32_t n_chunks[NSR_MAX_LEVELS];
    uint32_t slab_offset[NSR_MAX_LEVELS];  // floats, into the slab workspace (levels with n_chunks > 1)
    uint32_t entries_per_slice[NSR_MAX_LEVELS];
    uint32_t bin_offset[NSR_MAX_LEVELS];   // first (level, slice) bin of the level in the counter arrays
};

// LDS float atomics retire at ~0.33 lane-ops/clk/CU on gfx950, 64-bit INTEGER ones at ~5 (tools/lds_atomics_bench.hip),
// so the slices accumulate in Q27.36 fixed point: 15x the rate, 1.5e-11 resolution (the reference's tcnn accumulates
// this gradient in fp16), and -- integer addition being associative -- a bit-reproducible gradient on every level that is
// not split into chunk slabs (the slabs of the small dense levels are summed in fp32).
// t = value * 2^36 (|t| < 2^62) -> two's complement int64, built from exact fp32 pieces of |t|.
__device__ __forceinline__ unsigned long long own_to_fixed(float t)
{
    const float a = fabsf(t);
    const float th = floorf(a * 2.3283064365386963e-10f);  // floor(|t| / 2^32): the high word
    const float tl = rintf(fmaf(th, -4294967296.f, a));     // |t| - th * 2^32 in [0, 2^32): exact before the rint
    const unsigned long long v = ((unsigned long long)(uint32_t)th << 32) | (uint32_t)tl;
    return t < 0.f ? 0ull - v : v;
}

__device__ __forceinline__ float own_from_fixed(unsigned long long v) { return (float)(long long)v * OWN_FIX_INV; }

// An ITEM is one (y,z) corner pair of one sample on one level: the two x-neighbours share every hash/stride term and
// nearly always the owning slice.  word = sample << 4 | pair << 2 | mode, mode 0: both corners, 1: only x0, 2: only
// x0+1 (the pair straddles two slices: dense levels at a slice border or at the wrap-around of the last entries).
struct PairSlices { uint32_t s0, s1; };

__device__ __forceinline__ PairSlices pair_slices(const LevelGeom &g, const Cell &c, int k, uint32_t epb, bool pow2)
{
    const uint32_t cy = c.c[1] + (k & 1), cz = c.c[2] + (k >> 1);
    PairSlices p;
    if (pow2) {  // x never reaches the slice bits of a hashed level cut into 2^OWN_POW2_LOG2-entry slices
        const uint32_t h = (cy * PRIME_Y) ^ (cz * PRIME_Z);
        p.s0 = p.s1 = (h & (g.size - 1u)) >> OWN_POW2_LOG2;
    } else {
        p.s0 = corner_index(g, c.c[0], cy, cz) / epb;
        p.s1 = corner_index(g, c.c[0] + 1u, cy, cz) / epb;
    }
    return p;
}

__device__ __forceinline__ bool own_is_pow2(const LevelGeom &g, uint32_t epb)
{
    return !g.dense && epb == (1u << OWN_POW2_LOG2) && (g.size & (g.size - 1u)) == 0u && g.res < (1u << OWN_POW2_LOG2);
}


// ---- finite-difference stencils (neuralangelo, models/geometry.py:181-199): the table gradient of N samples x 7 points --
// The points of a sample are its position and six +-eps taps (layout [7][N][3], taps clamped to the box).  A tap that stays
// in the sample's cell of a level moves ONE coordinate inside a trilinear cell, so its corner weights are
// w(x) + delta * d w / d x_a exactly: everything those taps and the centre contribute to the cell's 8 corners is
//      w_c * G0  +  sum_a (d w_c / d x_a) * D_a ,    G0 = sum of their dy,  D_a = sum of delta * dy over the taps of axis a
// (delta in grid units, signed, the clamped taps' actual offsets).  Only taps that CROSS into a neighbouring cell keep
// items of their own.  At level 16 that is 16 merged + ~24 crossing point-items per sample instead of 112.
// cross[l][s]: bit t-1 set = tap t is in another cell than the centre on level l.
__global__ void __launch_bounds__(256)
k_tap_cross(const float *__restrict__ x7, uint32_t n_c, uint32_t mask_count, uint8_t *__restrict__ cross,
            const NsrGridDesc d)
{
    const uint32_t s = blockIdx.x * 256 + threadIdx.x, level = blockIdx.y;
    if (s >= n_c || level >= mask_count) return;
    const LevelGeom g = load_level(d, level);
    const Cell c0 = locate(g, x7[3ull * s], x7[3ull * s + 1], x7[3ull * s + 2]);
    uint32_t m = 0;
#pragma unroll
    for (int t = 1; t < 7; ++t) {
        const uint64_t i = (uint64_t)t * n_c + s;
        const Cell ct = locate(g, x7[3 * i], x7[3 * i + 1], x7[3 * i + 2]);
        if (ct.c[0] != c0.c[0] || ct.c[1] != c0.c[1] || ct.c[2] != c0.c[2]) m |= 1u << (t - 1);
    }
    cross[(uint64_t)level * n_c + s] = (uint8_t)m;
}

// G0 [L][N][F] and D [L][N][3][F] from the level-major dy of all 7N points ([L][7N][F])
template <int F>
__global__ void __launch_bounds__(256)
k_tap_reduce(const float *__restrict__ x7, const float *__restrict__ dy_lm, const uint8_t *__restrict__ cross,
             uint32_t n_c, uint32_t mask_count, float *__restrict__ g0, float *__restrict__ dd, const NsrGridDesc d)
{
    const uint32_t s = blockIdx.x * 256 + threadIdx.x, level = blockIdx.y;
    if (s >= n_c || level >= mask_count) return;
    const float scale = d.scale[level];
    const uint32_t m = cross[(uint64_t)level * n_c + s];
    const float *dyl = dy_lm + (uint64_t)level * 7ull * n_c * F;
    float G[F], D[3][F];
#pragma unroll
    for (int f = 0; f < F; ++f) { G[f] = dyl[(uint64_t)s * F + f]; D[0][f] = D[1][f] = D[2][f] = 0.f; }
#pragma unroll
    for (int t = 1; t < 7; ++t) {
        if (m & (1u << (t - 1))) continue;
        const int a = (t - 1) >> 1;
        const uint64_t i = (uint64_t)t * n_c + s;
        // same arithmetic as locate(): the offset of the tap inside the cell, in grid units
        const float delta = fmaf(scale, x7[3 * i + a], 0.5f) - fmaf(scale, x7[3ull * s + a], 0.5f);
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const float gv = dyl[i * F + f];
            G[f] += gv;
            D[a][f] = fmaf(delta, gv, D[a][f]);
        }
    }
    const uint64_t o = (uint64_t)level * n_c + s;
#pragma unroll
    for (int f = 0; f < F; ++f) {
        g0[o * F + f] = G[f];
#pragma unroll
        for (int a = 0; a < 3; ++a) dd[(o * 3 + a) * F + f] = D[a][f];
    }
}

// pass 1 (FILL = false): counts[bin] = number of items per (level, slice).
// pass 2 (FILL = true):  items[level][bin_start + ...] = item words; a block reserves one contiguous range per bin.
// grid (ceil(n / (OWN_BIN_BLOCK * OWN_BIN_SPT)), L)
template <bool FILL>
__global__ void __launch_bounds__(OWN_BIN_BLOCK)
k_own_bin(const float *__restrict__ x, uint32_t n, uint32_t mask_count, uint32_t *__restrict__ counts,
          const uint32_t *__restrict__ bin_start, uint32_t *__restrict__ cursors, uint32_t *__restrict__ items,
          const OwnerMap om, const NsrGridDesc d, const int32_t *__restrict__ n_dev,
          const uint8_t *__restrict__ cross /* stencil mode: tap points that stay in their sample's cell emit no items */,
          uint32_t n_c)
{
    __shared__ uint32_t hist[OWN_MAX_SLICES];
    const uint32_t level = blockIdx.y;
    if (level >= mask_count) return;
    const uint32_t n_live = live_count(n, n_dev);  // n stays the stride of the per-level item regions
    if (blockIdx.x * OWN_BIN_SPT * OWN_BIN_BLOCK >= n_live) return;
    const uint32_t R = om.n_slices[level], epb = om.entries_per_slice[level], bin0 = om.bin_offset[level];
    const LevelGeom g = load_level(d, level);
    const bool pow2 = own_is_pow2(g, epb);
    for (uint32_t s = threadIdx.x; s < R; s += OWN_BIN_BLOCK) hist[s] = 0u;
    __syncthreads();
    uint32_t slice[OWN_BIN_SPT][8], rank[OWN_BIN_SPT][8];
#pragma unroll
    for (int u = 0; u < OWN_BIN_SPT; ++u) {
        const uint32_t i = (blockIdx.x * OWN_BIN_SPT + u) * OWN_BIN_BLOCK + threadIdx.x;
        bool emit = i < n_live;
        if (emit && cross && i >= n_c) {
            const uint32_t t = i / n_c, sc = i - t * n_c;
            emit = (cross[(uint64_t)level * n_c + sc] >> (t - 1)) & 1u;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) slice[u][q] = 0xffffffffu;
        if (emit) {
            const Cell c = locate(g, x[3ull * i], x[3ull * i + 1], x[3ull * i + 2]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const PairSlices p = pair_slices(g, c, k, epb, pow2);
                // slot 2k: the pair (or its x0 half), slot 2k+1: the x0+1 half of a straddling pair
                slice[u][2 * k] = p.s0;
                rank[u][2 * k] = atomicAdd(&hist[p.s0], 1u);
                slice[u][2 * k + 1] = p.s1 != p.s0 ? p.s1 : 0xffffffffu;
                if (p.s1 != p.s0) rank[u][2 * k + 1] = atomicAdd(&hist[p.s1], 1u);
            }
        }
    }
    __syncthreads();
    if constexpr (!FILL) {
        for (uint32_t s = threadIdx.x; s < R; s += OWN_BIN_BLOCK)
            if (hist[s]) atomicAdd(&counts[bin0 + s], hist[s]);
    } else {
        // reserve this block's range in every bin it touches; hist[s] becomes the range's first item index
        for (uint32_t s = threadIdx.x; s < R; s += OWN_BIN_BLOCK)
            if (hist[s]) hist[s] = bin_start[bin0 + s] + atomicAdd(&cursors[bin0 + s], hist[s]);
        __syncthreads();
        uint32_t *dst = items + (uint64_t)level * n * 8ull;
#pragma unroll
        for (int u = 0; u < OWN_BIN_SPT; ++u) {
            const uint32_t i = (blockIdx.x * OWN_BIN_SPT + u) * OWN_BIN_BLOCK + threadIdx.x;
            if (i >= n_live) continue;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (slice[u][q] == 0xffffffffu) continue;  // (also: a tap point that emits nothing)
                const bool straddle = slice[u][q | 1] != 0xffffffffu;
                const uint32_t mode = !straddle ? 0u : ((q & 1) ? 2u : 1u);
                dst[hist[slice[u][q]] + rank[u][q]] = (i << 4) | ((uint32_t)(q >> 1) << 2) | mode;
            }
        }
    }
}

// per level: exclusive prefix of the bin counts -> bin_start; clears the fill cursors.  grid (L), block 256
__global__ void __launch_bounds__(256)
k_own_bin_scan(const uint32_t *__restrict__ counts, uint32_t *__restrict__ bin_start, uint32_t *__restrict__ cursors,
               const OwnerMap om)
{
    __shared__ uint32_t part[256];
    const uint32_t level = blockIdx.x, R = om.n_slices[level], bin0 = om.bin_offset[level];
    const uint32_t per = (R + 255u) / 256u, b = threadIdx.x * per, e = min(R, b + per);
    uint32_t sum = 0;
    for (uint32_t s = b; s < e; ++s) sum += counts[bin0 + s];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int t = 0; t < 256; ++t) { const uint32_t v = part[t]; part[t] = run; run += v; }
    }
    __syncthreads();
    uint32_t run = part[threadIdx.x];
    for (uint32_t s = b; s < e; ++s) {
        bin_start[bin0 + s] = run;
        run += counts[bin0 + s];
        cursors[bin0 + s] = 0u;
    }
}

template <int F>
__device__ __forceinline__ void lds_add(unsigned long long *acc, uint32_t rel, float w, const float (&g)[F])
{
#pragma unroll
    for (int f = 0; f < F; ++f) atomicAdd(&acc[rel * F + f], own_to_fixed(w * g[f]));
}

template <int F>
__device__ __forceinline__ void lds_add2(unsigned long long *acc, uint32_t rel, float w, const float (&g)[F], float w2,
                                         const float (&g2)[F])
{
#pragma unroll
    for (int f = 0; f < F; ++f) atomicAdd(&acc[rel * F + f], own_to_fixed(w * g[f] + w2 * g2[f]));
}

// Optional fused optimizer: the workgroup that owns a slice holds its finished gradient in LDS, so it applies AdamW to
// that slice right there (reads p / m / v, writes them + the fp16 image) instead of storing the gradient for a separate
// kernel to read back -- 8 B / parameter less HBM traffic and one 60 us kernel less on the step's critical path.
// Same arithmetic as csrc/util.hip (nsr_adamw_elem / nsr_adam_schedule): bit-identical parameters.
struct OwnerAdam {
    float *p, *m, *v;  // the TABLE slice of the parameter / moment vectors (entry 0 of level 0 first); NULL p: off
    __half *shadow;
    const int32_t *step;
    const float *hyper;
    double base_lr, b1d, b2d, gamma;
    int32_t m0, m1, m2;
    float b1, b2, eps, wd;
};

template <int F>
__device__ __forceinline__ void owner_adam4(const OwnerAdam &ad, uint64_t idx, const float (&gr)[4], float lr, float bc1,
                                            float bc2)
{
    float4 pp = *reinterpret_cast<const float4 *>(ad.p + idx), mm = *reinterpret_cast<const float4 *>(ad.m + idx);
    float4 vv = *reinterpret_cast<const float4 *>(ad.v + idx);
    float *pa = &pp.x, *ma = &mm.x, *va = &vv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) nsr_adamw_elem(pa[k], ma[k], va[k], gr[k], lr, ad.b1, ad.b2, ad.eps, ad.wd, bc1, bc2);
    *reinterpret_cast<float4 *>(ad.p + idx) = pp;
    *reinterpret_cast<float4 *>(ad.m + idx) = mm;
    *reinterpret_cast<float4 *>(ad.v + idx) = vv;
    if (ad.shadow) {
        __half2 h[2] = {__floats2half2_rn(pa[0], pa[1]), __floats2half2_rn(pa[2], pa[3])};
        *reinterpret_cast<uint2 *>(ad.shadow + idx) = *reinterpret_cast<uint2 *>(h);
    }
}

// Workgroup (level, slice, chunk): accumulates ITS items -- every lane busy, no scan over foreign samples.
// MODE 0: plain.  MODE 1: stencil mode (see k_tap_cross / k_tap_reduce): items of points < taps_nc carry the merged centre +
// in-cell taps.  MODE 2: second-order use (`dir` != NULL, optionally with the first-order term dy_first_lm of the same items).
// (Both are of the form  w_c G0 + sum_a (d w_c / d x_a) D_a  per corner -- mode 2 with G0 = dy_first, D_a = scale dir_a dy --
// but giving mode 2 the dense levels' run-length walk was measured slower: C3 2.96 -> 3.28 ms.)
template <int F, int MODE>
__global__ void __launch_bounds__(OWN_BLOCK)
k_grid_backward_owner(const float *__restrict__ x, const float *__restrict__ dy_lm /* [L][n][F] */,
                      const uint32_t *__restrict__ items, const uint32_t *__restrict__ counts,
                      const uint32_t *__restrict__ bin_start, float *__restrict__ grad_table,
                      float *__restrict__ slabs, uint32_t n, uint32_t mask_count, float grad_scale, int accumulate,
                      const OwnerMap om, const NsrGridDesc d, const float *__restrict__ dir,
                      const float *__restrict__ dy_first_lm /* with dir: first-order term of the SAME items, or NULL */,
                      const OwnerAdam ad, uint32_t taps_nc /* stencil mode: points < taps_nc are merged centre items */,
                      const float *__restrict__ tap_g0 /* [L][taps_nc][F] */, const float *__restrict__ tap_dd /* [L][taps_nc][3][F] */,
                      uint16_t *__restrict__ grad_bf16 /* non-NULL: the gradient leaves as bf16 (the multi-GPU transport buffer) */,
                      uint32_t row_base /* first block row of this launch: a launch may cover a run of levels only */)
{
    constexpr bool TAPS = MODE == 1;
    extern __shared__ __attribute__((aligned(16))) unsigned long long acc[];
    __shared__ float s_hyper[3];
    if (ad.p && threadIdx.x == 64) {  // a lane of the second wave: the schedule arithmetic (doubles) runs beside the LDS clear
        double p1, p2;
        nsr_adam_schedule(ad.step, ad.hyper, ad.base_lr, ad.b1d, ad.b2d, ad.gamma, ad.m0, ad.m1, ad.m2, s_hyper[0],
                          s_hyper[1], s_hyper[2], p1, p2);
    }
    __shared__ uint32_t s_nonfinite;  // an inf / NaN gradient reached this slice: it is flushed as NaN (GradScaler's
                                      // found_inf must fire exactly as it does with tcnn's fp16 atomics), never clamped away
    if (threadIdx.x == 0) s_nonfinite = 0u;
    const uint32_t xcd = blockIdx.x & 7u, j = (blockIdx.x >> 3) + row_base;
    uint32_t level = d.n_levels;
    uint32_t local = 0;
    for (uint32_t l = 0; l < d.n_levels; ++l) {
        const uint32_t wgs = om.n_slices[l] * om.n_chunks[l];
        const uint32_t t = (j - om.level_start[l]) * 8u + xcd;
        if (j >= om.level_start[l] && t < wgs) { level = l; local = t; }
    }
    if (level == d.n_levels) return;  // padding block of a lighter XCD
    const uint32_t C = om.n_chunks[level], epb = om.entries_per_slice[level];
    const uint32_t slice = local / C, chunk = local % C;
    const uint32_t r0 = slice * epb;
    const LevelGeom g = load_level(d, level);
    const uint32_t cnt = min(epb, g.size - r0);
    for (uint32_t k = threadIdx.x; k < cnt * F; k += OWN_BLOCK) acc[k] = 0ull;
    __syncthreads();
    if (level < mask_count) {
        const uint32_t bin = om.bin_offset[level] + slice;
        const uint32_t m = counts[bin];
        const uint32_t per = (m + C - 1) / C;
        const uint32_t i_beg = min(m, chunk * per), i_end = min(m, (chunk + 1) * per);
        const uint32_t *it = items + (uint64_t)level * n * 8ull + bin_start[bin];
        const float *dyl = dy_lm + (uint64_t)level * n * F;
        const float *dyf = dy_first_lm ? dy_first_lm + (uint64_t)level * n * F : nullptr;
        const float fix = grad_scale * OWN_FIX_SCALE;
        constexpr int OWN_BATCH = NSR_OWN_BATCH;  // items in flight per lane: item -> (x, dy) is a dependent load chain
        const float *g0l = TAPS ? tap_g0 + (uint64_t)level * taps_nc * F : nullptr;
        const float *ddl = TAPS ? tap_dd + (uint64_t)level * taps_nc * 3 * F : nullptr;
        // the run-length walk below hands every thread a CONTIGUOUS item range: with many items per thread the lanes of a
        // wave then read 64 different cache lines per load (measured: 1 M uniform samples 1.28 -> 1.90 ms), so it is used
        // up to NSR_OWN_RL_MAX items per thread only (the NeRF step has ~6); and not for the second-order mode's heavier items
        const uint32_t rl_q = (i_end - i_beg + OWN_BLOCK - 1) / OWN_BLOCK;
        if (g.dense && MODE != 2 && rl_q <= NSR_OWN_RL_MAX) {
            // Dense (coarse) levels: the binning passes lay the items of a slice down in runs of 64 CONSECUTIVE samples of
            // one corner pair, and consecutive samples of a ray sit in the same coarse cell for tens of steps -- handing a
            // wave 64 consecutive items makes its lanes hit the same two LDS words (64-way serialised atomics; DESIGN
            // section 7.2: levels 0-4 cost half of the kernel).  Here every THREAD walks a contiguous range of items and
            // keeps a run accumulator per corner in registers: one LDS atomic per (run, feature) instead of one per
            // (item, feature), and the lanes of a wave work on ranges that are far apart (different cells).
            const uint32_t m_c = i_end - i_beg, q = (m_c + OWN_BLOCK - 1) / OWN_BLOCK;
            const uint32_t j0 = min(i_end, i_beg + threadIdx.x * q), j1 = min(i_end, j0 + q);
            uint32_t key_lo = 0xffffffffu, key_hi = 0xffffffffu;
            float run_lo[F], run_hi[F];
#pragma unroll
            for (int f = 0; f < F; ++f) run_lo[f] = run_hi[f] = 0.f;
            for (uint32_t jb = j0; jb < j1; jb += OWN_BATCH) {
                uint32_t word[OWN_BATCH];
                float gb[OWN_BATCH][F], xb[OWN_BATCH][3], db[OWN_BATCH][3][F];
#pragma unroll
                for (int u = 0; u < OWN_BATCH; ++u) word[u] = jb + u < j1 ? it[jb + u] : 0xffffffffu;
#pragma unroll
                for (int u = 0; u < OWN_BATCH; ++u) {
                    const uint32_t s = word[u] != 0xffffffffu ? word[u] >> 4 : 0u;
                    const bool merged = TAPS && s < taps_nc;
#pragma unroll
                    for (int f = 0; f < F; ++f) {
                        gb[u][f] = merged ? g0l[(uint64_t)s * F + f] : dyl[(uint64_t)s * F + f];
#pragma unroll
                        for (int a = 0; a < 3; ++a) db[u][a][f] = merged ? ddl[((uint64_t)s * 3 + a) * F + f] : 0.f;
                    }
                    xb[u][0] = x[3ull * s]; xb[u][1] = x[3ull * s + 1]; xb[u][2] = x[3ull * s + 2];
                }
#pragma unroll
                for (int u = 0; u < OWN_BATCH; ++u) {
                    if (word[u] == 0xffffffffu) continue;
                    const int k = (word[u] >> 2) & 3;
                    const uint32_t mode = word[u] & 3u;
                    const Cell c = locate(g, xb[u][0], xb[u][1], xb[u][2]);
                    const uint32_t cy = c.c[1] + (k & 1), cz = c.c[2] + (k >> 1);
                    const float a1 = (k & 1) ? c.w[1] : 1.f - c.w[1], a2 = (k & 2) ? c.w[2] : 1.f - c.w[2];
                    const float a12 = a1 * a2;
                    const float w_lo = (1.f - c.w[0]) * a12, w_hi = c.w[0] * a12;
                    const uint32_t e_lo = mode != 2u ? corner_index(g, c.c[0], cy, cz) - r0 : 0xffffffffu;
                    const uint32_t e_hi = mode != 1u ? corner_index(g, c.c[0] + 1u, cy, cz) - r0 : 0xffffffffu;
                    // stencil mode, merged centre item: + sum_a (d w / d x_a) D_a  (zero D for plain items)
                    const float s1 = (k & 1) ? 1.f : -1.f, s2 = (k & 2) ? 1.f : -1.f;
                    const float dy_lo = (1.f - c.w[0]) * s1 * a2, dy_hi = c.w[0] * s1 * a2;
                    const float dz_lo = (1.f - c.w[0]) * a1 * s2, dz_hi = c.w[0] * a1 * s2;
                    float v_lo[F], v_hi[F];
#pragma unroll
                    for (int f = 0; f < F; ++f) {
                        if (!isfinite(gb[u][f])) s_nonfinite = 1u;
                        if constexpr (MODE == 1) {
                            if (!isfinite(db[u][0][f]) || !isfinite(db[u][1][f]) || !isfinite(db[u][2][f])) s_nonfinite = 1u;
                            const float t_lo = w_lo * gb[u][f] - a12 * db[u][0][f] + dy_lo * db[u][1][f] + dz_lo * db[u][2][f];
                            const float t_hi = w_hi * gb[u][f] + a12 * db[u][0][f] + dy_hi * db[u][1][f] + dz_hi * db[u][2][f];
                            v_lo[f] = fminf(fmaxf(t_lo * fix, -4.6e18f), 4.6e18f);
                            v_hi[f] = fminf(fmaxf(t_hi * fix, -4.6e18f), 4.6e18f);
                        } else {  // (the order of operations of the plain kernel: clamp(g * fix), then the weights)
                            const float v = fminf(fmaxf(gb[u][f] * fix, -4.6e18f), 4.6e18f);
                            v_lo[f] = w_lo * v;
                            v_hi[f] = w_hi * v;
                        }
                    }
                    if (e_lo != key_lo) {
                        if (key_lo != 0xffffffffu) {
#pragma unroll
                            for (int f = 0; f < F; ++f) atomicAdd(&acc[key_lo * F + f], own_to_fixed(fminf(fmaxf(run_lo[f], -4.6e18f), 4.6e18f)));
                        }
                        key_lo = e_lo;
#pragma unroll
                        for (int f = 0; f < F; ++f) run_lo[f] = 0.f;
                    }
                    if (e_hi != key_hi) {
                        if (key_hi != 0xffffffffu) {
#pragma unroll
                            for (int f = 0; f < F; ++f) atomicAdd(&acc[key_hi * F + f], own_to_fixed(fminf(fmaxf(run_hi[f], -4.6e18f), 4.6e18f)));
                        }
                        key_hi = e_hi;
#pragma unroll
                        for (int f = 0; f < F; ++f) run_hi[f] = 0.f;
                    }
#pragma unroll
                    for (int f = 0; f < F; ++f) { run_lo[f] += v_lo[f]; run_hi[f] += v_hi[f]; }
                }
            }
            if (key_lo != 0xffffffffu) {
#pragma unroll
                for (int f = 0; f < F; ++f) atomicAdd(&acc[key_lo * F + f], own_to_fixed(fminf(fmaxf(run_lo[f], -4.6e18f), 4.6e18f)));
            }
            if (key_hi != 0xffffffffu) {
#pragma unroll
                for (int f = 0; f < F; ++f) atomicAdd(&acc[key_hi * F + f], own_to_fixed(fminf(fmaxf(run_hi[f], -4.6e18f), 4.6e18f)));
            }
        } else
        for (uint32_t i0 = i_beg + threadIdx.x; i0 < i_end; i0 += OWN_BLOCK * OWN_BATCH) {
            uint32_t word[OWN_BATCH];
            float gb[OWN_BATCH][F], gf[OWN_BATCH][F], xb[OWN_BATCH][3];
#pragma unroll
            for (int u = 0; u < OWN_BATCH; ++u) {
                const uint32_t i = i0 + u * OWN_BLOCK;
                word[u] = i < i_end ? it[i] : 0xffffffffu;
            }
#pragma unroll
            for (int u = 0; u < OWN_BATCH; ++u) {
                const uint32_t s = word[u] != 0xffffffffu ? word[u] >> 4 : 0u;
                const float *src = (TAPS && s < taps_nc) ? g0l : dyl;
                if constexpr (F == 2) {
                    const float2 v = *reinterpret_cast<const float2 *>(src + 2ull * s);
                    gb[u][0] = v.x; gb[u][1] = v.y;
                } else {
#pragma unroll
                    for (int f = 0; f < F; ++f) gb[u][f] = src[(uint64_t)s * F + f];
                }
                xb[u][0] = x[3ull * s]; xb[u][1] = x[3ull * s + 1]; xb[u][2] = x[3ull * s + 2];
#pragma unroll
                for (int f = 0; f < F; ++f) gf[u][f] = dyf ? dyf[(uint64_t)s * F + f] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < OWN_BATCH; ++u) {
                if (word[u] == 0xffffffffu) continue;
                const int k = (word[u] >> 2) & 3;
                const uint32_t mode = word[u] & 3u;
                float g_out[F];
#pragma unroll
                for (int f = 0; f < F; ++f) {  // to fixed-point units; the clamp only keeps the integer conversion defined
                    if (!isfinite(gb[u][f])) s_nonfinite = 1u;
                    g_out[f] = fminf(fmaxf(gb[u][f] * fix, -4.6e18f), 4.6e18f);
                }
                const Cell c = locate(g, xb[u][0], xb[u][1], xb[u][2]);
                const uint32_t cy = c.c[1] + (k & 1), cz = c.c[2] + (k >> 1);
                const float a1 = (k & 1) ? c.w[1] : 1.f - c.w[1], a2 = (k & 2) ? c.w[2] : 1.f - c.w[2];
                float w_lo, w_hi;  // weights of the corners x0 and x0 + 1 of this (y,z) pair
                if (dir) {
                    // second-order use (double backward of the input gradient): the coefficient of table[corner] in
                    // sum_d dir_d * d(encoding)/dx_d  =  scale * sum_d dir_d * sign_d(corner) * prod_{e != d} w_e(corner)
                    const uint32_t smp = word[u] >> 4;
                    const float g0 = dir[3ull * smp], g1 = dir[3ull * smp + 1], g2 = dir[3ull * smp + 2];
                    const float s1 = (k & 1) ? 1.f : -1.f, s2 = (k & 2) ? 1.f : -1.f;
                    const float a0l = 1.f - c.w[0], a0h = c.w[0];
                    w_lo = g.scale * (-g0 * a1 * a2 + g1 * s1 * a0l * a2 + g2 * s2 * a0l * a1);
                    w_hi = g.scale * (g0 * a1 * a2 + g1 * s1 * a0h * a2 + g2 * s2 * a0h * a1);
                    if (!isfinite(w_lo) || !isfinite(w_hi)) { s_nonfinite = 1u; w_lo = w_hi = 0.f; }
                } else {
                    w_lo = (1.f - c.w[0]) * (a1 * a2);
                    w_hi = c.w[0] * (a1 * a2);
                }
                if constexpr (TAPS) {
                    const uint32_t smp = word[u] >> 4;
                    if (smp < taps_nc) {  // merged centre item: w G0 + sum_a (d w / d x_a) D_a, G0 was loaded as gb
                        const float s1 = (k & 1) ? 1.f : -1.f, s2 = (k & 2) ? 1.f : -1.f;
                        const float a12 = a1 * a2, a0l = 1.f - c.w[0], a0h = c.w[0];
                        float t_lo[F], t_hi[F];
#pragma unroll
                        for (int f = 0; f < F; ++f) {
                            const float dx = ddl[((uint64_t)smp * 3 + 0) * F + f], dyv = ddl[((uint64_t)smp * 3 + 1) * F + f],
                                        dz = ddl[((uint64_t)smp * 3 + 2) * F + f];
                            if (!isfinite(dx) || !isfinite(dyv) || !isfinite(dz)) s_nonfinite = 1u;
                            const float g_ = gb[u][f];
                            t_lo[f] = fminf(fmaxf((a0l * a12 * g_ - a12 * dx + a0l * s1 * a2 * dyv + a0l * a1 * s2 * dz) * fix, -4.6e18f), 4.6e18f);
                            t_hi[f] = fminf(fmaxf((a0h * a12 * g_ + a12 * dx + a0h * s1 * a2 * dyv + a0h * a1 * s2 * dz) * fix, -4.6e18f), 4.6e18f);
                        }
                        if (mode != 2u) lds_add<F>(acc, corner_index(g, c.c[0], cy, cz) - r0, 1.f, t_lo);
                        if (mode != 1u) lds_add<F>(acc, corner_index(g, c.c[0] + 1u, cy, cz) - r0, 1.f, t_hi);
                        continue;
                    }
                }
                if (dyf) {
                    // first-order and second-order terms of one training step share their items (same samples, same
                    // corners): w_dir * dy_second + w_trilinear * dy_first leaves as ONE fixed-point atomic per entry
                    float f_out[F];
#pragma unroll
                    for (int f = 0; f < F; ++f) {
                        if (!isfinite(gf[u][f])) s_nonfinite = 1u;
                        f_out[f] = fminf(fmaxf(gf[u][f] * fix, -4.6e18f), 4.6e18f);
                    }
                    const float f_lo = (1.f - c.w[0]) * (a1 * a2), f_hi = c.w[0] * (a1 * a2);
                    if (mode != 2u) lds_add2<F>(acc, corner_index(g, c.c[0], cy, cz) - r0, w_lo, g_out, f_lo, f_out);
                    if (mode != 1u) lds_add2<F>(acc, corner_index(g, c.c[0] + 1u, cy, cz) - r0, w_hi, g_out, f_hi, f_out);
                    continue;
                }
                if (mode != 2u) lds_add<F>(acc, corner_index(g, c.c[0], cy, cz) - r0, w_lo, g_out);
                if (mode != 1u) lds_add<F>(acc, corner_index(g, c.c[0] + 1u, cy, cz) - r0, w_hi, g_out);
            }
        }
    }
    __syncthreads();
    const uint32_t nf = cnt * F;  // multiple of 8: level sizes are multiples of 8 entries
    if (ad.p && C == 1) {  // fused AdamW on the slice this workgroup owns (a non-finite slice updates with NaN gradients)
        const float lr = s_hyper[0], bc1 = s_hyper[1], bc2 = s_hyper[2];
        const uint64_t base = (uint64_t)(g.offset + r0) * F;
        const bool bad = s_nonfinite != 0u;
        const float qnan = __builtin_nanf("");
        // every load of the slice's p / m / v is issued before the first store: the pointers may alias as far as the
        // compiler knows, so a load / compute / store loop would pay one memory round trip per iteration
        constexpr int ADAM_IT = (OWN_LDS_WORDS + OWN_BLOCK * 4 - 1) / (OWN_BLOCK * 4);
        float4 pp[ADAM_IT], mm[ADAM_IT], vv[ADAM_IT];
#pragma unroll
        for (int it = 0; it < ADAM_IT; ++it) {
            const uint32_t k = (it * OWN_BLOCK + threadIdx.x) * 4;
            if (k < nf) {
                pp[it] = *reinterpret_cast<const float4 *>(ad.p + base + k);
                mm[it] = *reinterpret_cast<const float4 *>(ad.m + base + k);
                vv[it] = *reinterpret_cast<const float4 *>(ad.v + base + k);
            }
        }
#pragma unroll
        for (int it = 0; it < ADAM_IT; ++it) {
            const uint32_t k = (it * OWN_BLOCK + threadIdx.x) * 4;
            if (k >= nf) continue;
            float *pa = &pp[it].x, *ma = &mm[it].x, *va = &vv[it].x;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                nsr_adamw_elem(pa[q], ma[q], va[q], bad ? qnan : own_from_fixed(acc[k + q]), lr, ad.b1, ad.b2, ad.eps, ad.wd,
                               bc1, bc2);
            *reinterpret_cast<float4 *>(ad.p + base + k) = pp[it];
            *reinterpret_cast<float4 *>(ad.m + base + k) = mm[it];
            *reinterpret_cast<float4 *>(ad.v + base + k) = vv[it];
            if (ad.shadow) {
                __half2 h[2] = {__floats2half2_rn(pa[0], pa[1]), __floats2half2_rn(pa[2], pa[3])};
                *reinterpret_cast<uint2 *>(ad.shadow + base + k) = *reinterpret_cast<uint2 *>(h);
            }
        }
        return;
    }
    if (s_nonfinite) {
        const float qnan = __builtin_nanf("");
        if (grad_bf16 && C == 1) {
            uint16_t *dst = grad_bf16 + (uint64_t)(g.offset + r0) * F;
            for (uint32_t k = threadIdx.x; k < nf; k += OWN_BLOCK) dst[k] = 0x7fc0u;
            return;
        }
        float *dst = C > 1 ? slabs + om.slab_offset[level] + ((uint64_t)chunk * g.size + r0) * F
                           : grad_table + (uint64_t)(g.offset + r0) * F;
        for (uint32_t k = threadIdx.x; k < nf; k += OWN_BLOCK) dst[k] = qnan;
        return;
    }
    if (C > 1) {
        float *dst = slabs + om.slab_offset[level] + ((uint64_t)chunk * g.size + r0) * F;
        for (uint32_t k = threadIdx.x * 4; k < nf; k += OWN_BLOCK * 4)
            *reinterpret_cast<float4 *>(dst + k) = make_float4(own_from_fixed(acc[k]), own_from_fixed(acc[k + 1]),
                                                               own_from_fixed(acc[k + 2]), own_from_fixed(acc[k + 3]));
    } else if (grad_bf16) {  // transport format of the multi-GPU exchange: written once, never accumulated into
        uint16_t *dst = grad_bf16 + (uint64_t)(g.offset + r0) * F;
        for (uint32_t k = threadIdx.x * 4; k < nf; k += OWN_BLOCK * 4) {
            uint2 o;
            o.x = nsr_pack_bf16x2(own_from_fixed(acc[k]), own_from_fixed(acc[k + 1]));
            o.y = nsr_pack_bf16x2(own_from_fixed(acc[k + 2]), own_from_fixed(acc[k + 3]));
            *reinterpret_cast<uint2 *>(dst + k) = o;
        }
    } else {
        float *dst = grad_table + (uint64_t)(g.offset + r0) * F;
        for (uint32_t k = threadIdx.x * 4; k < nf; k += OWN_BLOCK * 4) {
            float4 v = make_float4(own_from_fixed(acc[k]), own_from_fixed(acc[k + 1]), own_from_fixed(acc[k + 2]),
                                   own_from_fixed(acc[k + 3]));
            if (accumulate) {
                const float4 o = *reinterpret_cast<const float4 *>(dst + k);
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
            *reinterpret_cast<float4 *>(dst + k) = v;
        }
    }
}

// grad[level] (+)= sum over that level's chunk slabs (levels with n_chunks > 1 only)
template <int F>
__global__ void __launch_bounds__(256)
k_grid_reduce_slabs(const float *__restrict__ slabs, float *__restrict__ grad_table, int accumulate, const OwnerMap om,
                    const NsrGridDesc d, const OwnerAdam ad, uint16_t *__restrict__ grad_bf16, uint32_t level_base)
{
    const uint32_t level = blockIdx.y + level_base;
    const uint32_t C = om.n_chunks[level];
    if (C <= 1) return;
    __shared__ float s_hyper[3];
    if (ad.p) {
        if (threadIdx.x == 0) {
            double p1, p2;
            nsr_adam_schedule(ad.step, ad.hyper, ad.base_lr, ad.b1d, ad.b2d, ad.gamma, ad.m0, ad.m1, ad.m2, s_hyper[0],
                              s_hyper[1], s_hyper[2], p1, p2);
        }
        __syncthreads();
    }
    const uint32_t nf = d.size[level] * F;
    const float *src = slabs + om.slab_offset[level];
    float *dst = grad_table + (uint64_t)d.offset[level] * F;
    // levels split into many chunks are small (level 0: 8 K floats x 64 slabs): one thread per float4 would leave 2,048
    // threads walking 64 slabs each.  S lanes share a float4, each sums every S-th slab, a shuffle tree joins them
    // (fixed order: the result does not depend on the launch).
    const uint32_t S = C >= 32 ? 8u : (C >= 8 ? 4u : 1u);
    const uint32_t t = blockIdx.x * 256 + threadIdx.x, sub = t % S, q0 = t / S, qs = gridDim.x * 256 / S;
    for (uint32_t k = q0 * 4; k < nf; k += qs * 4) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (uint32_t c = sub; c < C; c += S) {  // unrolled: independent loads in flight
            const float4 v = *reinterpret_cast<const float4 *>(src + (uint64_t)c * nf + k);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        for (uint32_t o = S >> 1; o > 0; o >>= 1) {
            s.x += __shfl_xor(s.x, o, 64); s.y += __shfl_xor(s.y, o, 64);
            s.z += __shfl_xor(s.z, o, 64); s.w += __shfl_xor(s.w, o, 64);
        }
        if (sub != 0) continue;
        if (accumulate) {
            const float4 o4 = *reinterpret_cast<const float4 *>(dst + k);
            s.x += o4.x; s.y += o4.y; s.z += o4.z; s.w += o4.w;
        }
        if (ad.p) {  // the summed gradient of a dense level goes straight into AdamW (see OwnerAdam)
            const float gr[4] = {s.x, s.y, s.z, s.w};
            owner_adam4<F>(ad, (uint64_t)d.offset[level] * F + k, gr, s_hyper[0], s_hyper[1], s_hyper[2]);
        } else if (grad_bf16) {
            uint2 o;
            o.x = nsr_pack_bf16x2(s.x, s.y);
            o.y = nsr_pack_bf16x2(s.z, s.w);
            *reinterpret_cast<uint2 *>(grad_bf16 + (uint64_t)d.offset[level] * F + k) = o;
        } else {
            *reinterpret_cast<float4 *>(dst + k) = s;
        }
    }
}

// host: build the decomposition; returns the number of blocks, *slab_floats the slab workspace size, *n_bins the number
// of (level, slice) bins
static uint32_t make_owner_map(const NsrGridDesc *desc, OwnerMap *om, uint64_t *slab_floats, uint32_t *n_bins)
{
    const uint32_t F = desc->n_features, L = desc->n_levels;
    uint64_t slab = 0;
    uint32_t bins = 0, rows = 0;
    for (uint32_t l = 0; l < NSR_MAX_LEVELS; ++l)
        om->level_start[l] = om->n_slices[l] = om->n_chunks[l] = om->slab_offset[l] = om->entries_per_slice[l] =
            om->bin_offset[l] = 0;
    for (uint32_t l = 0; l < L; ++l) {
        const uint32_t size = desc->size[l], res = desc->resolution[l];
        const bool dense = (uint64_t)res * res * res <= (uint64_t)size;
        const uint32_t max_epb = OWN_LDS_WORDS / F;
        uint32_t epb = max_epb;
        const uint32_t p2 = 1u << OWN_POW2_LOG2;
        if (!dense && (size & (size - 1)) == 0 && size >= p2 && p2 <= max_epb && res < p2) epb = p2;
        const uint32_t R = nsr_div_up(size, epb);
        const uint32_t target = dense ? OWN_DENSE_TARGET_WGS : OWN_TARGET_WGS;
        uint32_t C = 1;
        if (R < target) C = (target + R - 1) / R;  // few slices: split the items instead
        om->n_slices[l] = R;
        om->n_chunks[l] = C;
        om->entries_per_slice[l] = epb;
        om->slab_offset[l] = (uint32_t)slab;
        om->bin_offset[l] = bins;
        bins += R;
        if (C > 1) slab += (uint64_t)C * size * F;
        om->level_start[l] = rows;
        rows += nsr_div_up(R * C, 8);
    }
    *slab_floats = slab;
    *n_bins = bins;
    return rows * 8;
}

// row-major dy [n, stride] (half or float) -> level-major fp32 [L][n][F]; 64 samples x all columns per block
template <int F, bool DY_F32>
__global__ void __launch_bounds__(256)
k_dy_to_level_major(const void *__restrict__ dy, uint32_t dy_stride, float *__restrict__ out, uint32_t n, uint32_t L)
{
    extern __shared__ float tile[];  // [64][C+1]
    const uint32_t C = L * F, i0 = blockIdx.x * 64;
    for (uint32_t k = threadIdx.x; k < 64 * C; k += 256) {
        const uint32_t r = k / C, c = k % C;
        tile[r * (C + 1) + c] = (i0 + r < n) ? load_grad<DY_F32>(dy, (uint64_t)(i0 + r) * dy_stride + c) : 0.f;
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < 64 * C; k += 256) {
        const uint32_t l = k / (64 * F), rem = k % (64 * F), r = rem / F, f = rem % F;
        if (i0 + r < n) out[((uint64_t)l * n + i0 + r) * F + f] = tile[r * (C + 1) + l * F + f];
    }
}

// ------------------------------------------------------------------------------------------------
// backward w.r.t. the input (and its double backward): one lane = one sample, loop over levels.
// dy/dx is recomputed from the (cache-resident) table instead of being stored by the forward pass.
// ------------------------------------------------------------------------------------------------
template <int F, bool DY_F32>
__global__ void __launch_bounds__(GRID_BLOCK)
k_grid_backward_input(const float *__restrict__ x, const __half *__restrict__ table, const void *__restrict__ dy,
                      uint32_t dy_stride, float *__restrict__ dx, uint32_t n, uint32_t mask_count,
                      const NsrGridDesc d)
{
    const uint32_t i = blockIdx.x * GRID_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float x0 = x[3ull * i], x1 = x[3ull * i + 1], x2 = x[3ull * i + 2];
    float gx[3] = {0.f, 0.f, 0.f};
    const uint32_t nl = min(d.n_levels, mask_count);
    for (uint32_t level = 0; level < nl; ++level) {
        const LevelGeom g = load_level(d, level);
        const Cell c = locate(g, x0, x1, x2);
        float gy[F];
#pragma unroll
        for (int f = 0; f < F; ++f) gy[f] = load_grad<DY_F32>(dy, (uint64_t)i * dy_stride + level * F + f);
        // s[k] = sum_f dy_f * table[corner k][f]
        float s[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t e = corner_index(g, c.c[0] + (k & 1), c.c[1] + ((k >> 1) & 1), c.c[2] + ((k >> 2) & 1));
            float v[F];
            load_feat<F>(table, g.offset + e, v);
            float a = 0.f;
#pragma unroll
            for (int f = 0; f < F; ++f) a = fmaf(gy[f], v[f], a);
            s[k] = a;
        }
        const float w0 = c.w[0], w1 = c.w[1], w2 = c.w[2];
        // d/dx0: corners differ in bit0; weight = w1(bit1) * w2(bit2)
        const float d0 = (1.f - w1) * (1.f - w2) * (s[1] - s[0]) + w1 * (1.f - w2) * (s[3] - s[2]) +
                         (1.f - w1) * w2 * (s[5] - s[4]) + w1 * w2 * (s[7] - s[6]);
        const float d1 = (1.f - w0) * (1.f - w2) * (s[2] - s[0]) + w0 * (1.f - w2) * (s[3] - s[1]) +
                         (1.f - w0) * w2 * (s[6] - s[4]) + w0 * w2 * (s[7] - s[5]);
        const float d2 = (1.f - w0) * (1.f - w1) * (s[4] - s[0]) + w0 * (1.f - w1) * (s[5] - s[1]) +
                         (1.f - w0) * w1 * (s[6] - s[2]) + w0 * w1 * (s[7] - s[3]);
        gx[0] = fmaf(g.scale, d0, gx[0]);
        gx[1] = fmaf(g.scale, d1, gx[1]);
        gx[2] = fmaf(g.scale, d2, gx[2]);
    }
    dx[3ull * i] = gx[0]; dx[3ull * i + 1] = gx[1]; dx[3ull * i + 2] = gx[2];
}

// Double backward of  dx = J(x; table)^T dy  given g = dL/d(dx):
//   d_dy[l,f]            = sum_d g_d * dJ_{lf,d}
//   grad_table[corner,f] += dy_lf * scale * sum_d g_d * sign_d(corner) * prod_{e!=d} w_e(corner)
//   dx2_e                = sum_{d!=e} g_d * sum_lf dy_lf * scale^2 * sum_corner sign_d sign_e w_third T[corner,f]
template <int F, bool DY_F32>
__global__ void __launch_bounds__(GRID_BLOCK)
k_grid_bwd_bwd_input(const float *__restrict__ x, const __half *__restrict__ table, const void *__restrict__ dy,
                     uint32_t dy_stride, const float *__restrict__ gin, float *__restrict__ d_dy,
                     uint32_t d_dy_stride, float *__restrict__ grad_table, float *__restrict__ dx2, uint32_t n,
                     uint32_t mask_count, const NsrGridDesc d)
{
    const uint32_t i = blockIdx.x * GRID_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float x0 = x[3ull * i], x1 = x[3ull * i + 1], x2 = x[3ull * i + 2];
    const float g0 = gin[3ull * i], g1 = gin[3ull * i + 1], g2 = gin[3ull * i + 2];
    float acc2[3] = {0.f, 0.f, 0.f};
    const uint32_t nl = min(d.n_levels, mask_count);
    for (uint32_t level = 0; level < d.n_levels; ++level) {
        if (level >= nl) {
            if (d_dy)
                for (int f = 0; f < F; ++f) d_dy[(uint64_t)i * d_dy_stride + level * F + f] = 0.f;
            continue;
        }
        const LevelGeom g = load_level(d, level);
        const Cell c = locate(g, x0, x1, x2);
        const float w[3] = {c.w[0], c.w[1], c.w[2]};
        float gy[F];
#pragma unroll
        for (int f = 0; f < F; ++f) gy[f] = load_grad<DY_F32>(dy, (uint64_t)i * dy_stride + level * F + f);
        float ddy[F];
#pragma unroll
        for (int f = 0; f < F; ++f) ddy[f] = 0.f;
        float m01 = 0.f, m02 = 0.f, m12 = 0.f;  // mixed second derivatives contracted with dy
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t e = corner_index(g, c.c[0] + (k & 1), c.c[1] + ((k >> 1) & 1), c.c[2] + ((k >> 2) & 1));
            float v[F];
            load_feat<F>(table, g.offset + e, v);
            const float a0 = (k & 1) ? w[0] : 1.f - w[0], a1 = (k & 2) ? w[1] : 1.f - w[1],
                        a2 = (k & 4) ? w[2] : 1.f - w[2];
            const float s0 = (k & 1) ? 1.f : -1.f, s1 = (k & 2) ? 1.f : -1.f, s2 = (k & 4) ? 1.f : -1.f;
            // coefficient of table[corner] in  sum_d g_d * dy/dx_d  (without scale)
            const float coef = g0 * s0 * a1 * a2 + g1 * s1 * a0 * a2 + g2 * s2 * a0 * a1;
            float sv = 0.f;
#pragma unroll
            for (int f = 0; f < F; ++f) {
                ddy[f] = fmaf(coef, v[f], ddy[f]);
                sv = fmaf(gy[f], v[f], sv);
            }
            if (grad_table) {
                float *gp = grad_table + (uint64_t)(g.offset + e) * F;
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    const float t = g.scale * coef * gy[f];
                    if (t != 0.f) unsafeAtomicAdd(gp + f, t);
                }
            }
            m01 = fmaf(s0 * s1 * a2, sv, m01);
            m02 = fmaf(s0 * s2 * a1, sv, m02);
            m12 = fmaf(s1 * s2 * a0, sv, m12);
        }
        if (d_dy) {
#pragma unroll
            for (int f = 0; f < F; ++f) d_dy[(uint64_t)i * d_dy_stride + level * F + f] = g.scale * ddy[f];
        }
        const float sc2 = g.scale * g.scale;
        acc2[0] = fmaf(sc2, g1 * m01 + g2 * m02, acc2[0]);
        acc2[1] = fmaf(sc2, g0 * m01 + g2 * m12, acc2[1]);
        acc2[2] = fmaf(sc2, g0 * m02 + g1 * m12, acc2[2]);
    }
    if (dx2) { dx2[3ull * i] = acc2[0]; dx2[3ull * i + 1] = acc2[1]; dx2[3ull * i + 2] = acc2[2]; }
}

int check_desc(const NsrGridDesc *d, const char *who)
{
    NSR_REQUIRE(d != nullptr, "%s: desc is NULL", who);
    NSR_REQUIRE(d->n_levels >= 1 && d->n_levels <= NSR_MAX_LEVELS, "%s: n_levels=%u out of range", who, d->n_levels);
    NSR_REQUIRE(d->n_features == 1 || d->n_features == 2 || d->n_features == 4 || d->n_features == 8,
                "%s: n_features=%u unsupported (1,2,4,8)", who, d->n_features);
    return NSR_OK;
}

}  // namespace

#define DISPATCH_F(F_, ...)                  \
    switch (F_) {                            \
    case 1: { constexpr int F = 1; __VA_ARGS__; } break; \
    case 2: { constexpr int F = 2; __VA_ARGS__; } break; \
    case 4: { constexpr int F = 4; __VA_ARGS__; } break; \
    default: { constexpr int F = 8; __VA_ARGS__; } break; \
    }

extern "C" int nsr_hashgrid_make_desc(NsrGridDesc *out, uint32_t n_levels, uint32_t n_features,
                                      uint32_t log2_hashmap_size, uint32_t base_resolution, float per_level_scale)
{
    NSR_REQUIRE(out != nullptr, "nsr_hashgrid_make_desc: out is NULL");
    NSR_REQUIRE(n_levels >= 1 && n_levels <= NSR_MAX_LEVELS, "nsr_hashgrid_make_desc: n_levels=%u", n_levels);
    NSR_REQUIRE(log2_hashmap_size >= 3 && log2_hashmap_size <= 28, "nsr_hashgrid_make_desc: log2_hashmap_size=%u",
                log2_hashmap_size);
    memset(out, 0, sizeof(*out));
    out->n_levels = n_levels;
    out->n_features = n_features;
    out->log2_hashmap_size = log2_hashmap_size;
    out->base_resolution = base_resolution;
    out->per_level_scale = per_level_scale;
    const float log2s = log2f(per_level_scale);
    uint32_t off = 0;
    for (uint32_t l = 0; l < n_levels; ++l) {
        volatile float a = (float)l * log2s;  // every op separately rounded to fp32
        volatile float e = exp2f(a);
        volatile float m = e * (float)base_resolution;
        const float scale = m - 1.0f;
        const uint32_t res = (uint32_t)ceilf(scale) + 1u;
        uint64_t cells = (uint64_t)res * res * res;
        if (cells > 0xFFFFFFFFull) cells = 0xFFFFFFFFull;
        cells = (cells + 7ull) / 8ull * 8ull;
        const uint64_t cap = 1ull << log2_hashmap_size;
        const uint32_t size = (uint32_t)(cells < cap ? cells : cap);
        out->scale[l] = scale;
        out->resolution[l] = res;
        out->size[l] = size;
        out->offset[l] = off;
        off += size;
    }
    out->offset[n_levels] = off;
    out->n_entries = off;
    return check_desc(out, "nsr_hashgrid_make_desc");
}

extern "C" int nsr_hashgrid_forward_ex(const float *x, const nsr_half *table, nsr_half *y, uint32_t n, uint32_t y_stride,
                                       int y_level_major, uint32_t level_mask_count, const NsrGridDesc *desc,
                                       const int32_t *n_dev, void *stream);

// default (0, 2): measured 22 % faster than (0, 1) on ray-coherent samples, (1..2, *) slower -- profiles/r02_forward_ab.json
static int g_fwd_lds_levels = 0, g_fwd_levels_per_lane = 2;
extern "C" int nsr_hashgrid_forward_variant(int lds_levels, int levels_per_lane)
{
    NSR_REQUIRE(lds_levels >= 0 && lds_levels <= 4 && (levels_per_lane == 1 || levels_per_lane == 2),
                "nsr_hashgrid_forward_variant: lds_levels 0..4, levels_per_lane 1 or 2");
    g_fwd_lds_levels = lds_levels;
    g_fwd_levels_per_lane = levels_per_lane;
    return NSR_OK;
}

extern "C" int nsr_hashgrid_forward(const float *x, const nsr_half *table, nsr_half *y, uint32_t n, uint32_t y_stride,
                                    uint32_t level_mask_count, const NsrGridDesc *desc, void *stream)
{
    return nsr_hashgrid_forward_ex(x, table, y, n, y_stride, 0, level_mask_count, desc, nullptr, stream);
}

extern "C" int nsr_hashgrid_forward_ex(const float *x, const nsr_half *table, nsr_half *y, uint32_t n, uint32_t y_stride,
                                       int y_level_major, uint32_t level_mask_count, const NsrGridDesc *desc,
                                       const int32_t *n_dev, void *stream)
{
    if (int rc = check_desc(desc, "nsr_hashgrid_forward")) return rc;
    NSR_REQUIRE(y_level_major || y_stride >= desc->n_levels * desc->n_features, "nsr_hashgrid_forward: y_stride too small");
    if (n == 0) return NSR_OK;
    NSR_REQUIRE(x && table && y, "nsr_hashgrid_forward: NULL pointer");
    const uint32_t lpx = (desc->n_levels + 7) / 8;
    const uint32_t grid = 8u * lpx * nsr_div_up(n, GRID_BLOCK);
    uint32_t level_begin = 0;
    if (g_fwd_lds_levels > 0) {  // A/B variant: leading small dense levels from LDS
        while (level_begin < (uint32_t)g_fwd_lds_levels && level_begin < desc->n_levels &&
               desc->size[level_begin] <= 16384u &&
               (uint64_t)desc->resolution[level_begin] * desc->resolution[level_begin] * desc->resolution[level_begin] <=
                   desc->size[level_begin])
            ++level_begin;
        if (level_begin > 0) {
            const uint32_t bpl = 128;
            uint32_t max_size = 0;
            for (uint32_t l = 0; l < level_begin; ++l) max_size = desc->size[l] > max_size ? desc->size[l] : max_size;
            const size_t lds = (size_t)max_size * desc->n_features * 2;
            DISPATCH_F(desc->n_features, {
                (void)hipFuncSetAttribute((const void *)k_grid_forward_lds<F>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                hipLaunchKernelGGL((k_grid_forward_lds<F>), dim3(bpl * level_begin), dim3(LDS_FWD_BLOCK), lds,
                                   (hipStream_t)stream, x, (const __half *)table, (__half *)y, n, y_stride,
                                   level_mask_count, y_level_major, bpl, *desc, n_dev);
            });
        }
    }
    if (g_fwd_levels_per_lane == 2 && desc->n_levels <= 16) {
        DISPATCH_F(desc->n_features,
                   hipLaunchKernelGGL((k_grid_forward_pair<F>), dim3(8u * nsr_div_up(n, GRID_BLOCK)), dim3(GRID_BLOCK), 0,
                                      (hipStream_t)stream, x, (const __half *)table, (__half *)y, n, y_stride,
                                      level_mask_count, level_begin, y_level_major, *desc, n_dev));
        NSR_CHECK_LAUNCH("nsr_hashgrid_forward(pair)");
        return NSR_OK;
    }
    DISPATCH_F(desc->n_features,
               hipLaunchKernelGGL((k_grid_forward<F>), dim3(grid), dim3(GRID_BLOCK), 0, (hipStream_t)stream, x,
                                  (const __half *)table, (__half *)y, n, y_stride, level_mask_count, lpx, y_level_major,
                                  *desc, n_dev, (float *)nullptr, level_begin));
    NSR_CHECK_LAUNCH("nsr_hashgrid_forward");
    return NSR_OK;
}

extern "C" int nsr_hashgrid_forward_jac(const float *x, const nsr_half *table, nsr_half *y, uint32_t n, uint32_t y_stride,
                                        int y_level_major, uint32_t level_mask_count, const NsrGridDesc *desc, float *jac,
                                        const int32_t *n_dev, void *stream)
{
    if (int rc = check_desc(desc, "nsr_hashgrid_forward_jac")) return rc;
    NSR_REQUIRE(y_level_major || y_stride >= desc->n_levels * desc->n_features, "nsr_hashgrid_forward_jac: y_stride too small");
    if (n == 0) return NSR_OK;
    NSR_REQUIRE(x && table && y && jac, "nsr_hashgrid_forward_jac: NULL pointer");
    const uint32_t lpx = (desc->n_levels + 7) / 8;
    const uint32_t grid = 8u * lpx * nsr_div_up(n, GRID_BLOCK);
    DISPATCH_F(desc->n_features,
               hipLaunchKernelGGL((k_grid_forward<F>), dim3(grid), dim3(GRID_BLOCK), 0, (hipStream_t)stream, x,
                                  (const __half *)table, (__half *)y, n, y_stride, level_mask_count, lpx, y_level_major,
                                  *desc, n_dev, jac, 0u));
    NSR_CHECK_LAUNCH("nsr_hashgrid_forward_jac");
    return NSR_OK;
}

namespace {
// dx[i][:] = sum_c dy[i][c] J[c][i][:]   and / or   d_dy[i][c] = J[c][i][:] . g[i][:]     (J level-major [L][n][F][3])
// dy / d_dy are ROW-major (what the MFMA kernels read and write: 144-B rows): a block moves its 256 x C tile through
// LDS with coalesced row segments instead of letting every lane walk its own row (64 cache lines per load instruction)
constexpr int JAC_BLOCK = 256;
__global__ void __launch_bounds__(JAC_BLOCK)
k_jac_apply(const float *__restrict__ jac, uint32_t n, uint32_t L, uint32_t F, const float *__restrict__ dy,
            uint32_t dy_stride, float *__restrict__ dx, const float *__restrict__ g, float *__restrict__ d_dy,
            uint32_t d_dy_stride, const int32_t *__restrict__ n_dev)
{
    extern __shared__ float tile[];  // [JAC_BLOCK][C + 1]
    const uint32_t C = L * F, ld = C + 1;
    const uint32_t n_live = live_count(n, n_dev);
    const uint32_t i0 = blockIdx.x * JAC_BLOCK, i = i0 + threadIdx.x;
    if (i0 >= n_live) return;
    const uint32_t rows = min((uint32_t)JAC_BLOCK, n_live - i0);
    if (dx) {
        for (uint32_t k = threadIdx.x; k < rows * C; k += JAC_BLOCK)
            tile[(k / C) * ld + k % C] = dy[(uint64_t)(i0 + k / C) * dy_stride + k % C];
        __syncthreads();
    }
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, g0 = 0.f, g1 = 0.f, g2 = 0.f;
    const bool live = i < n_live;
    if (live && g) { g0 = g[3ull * i]; g1 = g[3ull * i + 1]; g2 = g[3ull * i + 2]; }
    if (live) {
        for (uint32_t l = 0; l < L; ++l) {
            const float *j = jac + ((uint64_t)l * n + i) * (F * 3);
            for (uint32_t f = 0; f < F; ++f) {
                const float j0 = j[f * 3], j1 = j[f * 3 + 1], j2 = j[f * 3 + 2];
                float *t = &tile[threadIdx.x * ld + l * F + f];
                if (dx) {
                    const float w = *t;
                    a0 = fmaf(w, j0, a0); a1 = fmaf(w, j1, a1); a2 = fmaf(w, j2, a2);
                }
                if (d_dy) *t = j0 * g0 + j1 * g1 + j2 * g2;  // (after the read above: same thread, same slot)
            }
        }
        if (dx) { dx[3ull * i] = a0; dx[3ull * i + 1] = a1; dx[3ull * i + 2] = a2; }
    }
    if (d_dy) {
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < rows * C; k += JAC_BLOCK)
            d_dy[(uint64_t)(i0 + k / C) * d_dy_stride + k % C] = tile[(k / C) * ld + k % C];
    }
}
}  // namespace

extern "C" int nsr_hashgrid_jac_apply(const float *jac, uint32_t n, const NsrGridDesc *desc, const float *dy,
                                      uint32_t dy_stride, float *dx, const float *g, float *d_dy, uint32_t d_dy_stride,
                                      const int32_t *n_dev, void *stream)
{
    if (int rc = check_desc(desc, "nsr_hashgrid_jac_apply")) return rc;
    if (n == 0) return NSR_OK;
    NSR_REQUIRE(jac && ((dx && dy) || (d_dy && g)), "nsr_hashgrid_jac_apply: NULL pointer");
    const size_t lds = (size_t)JAC_BLOCK * (desc->n_levels * desc->n_features + 1) * sizeof(float);
    hipLaunchKernelGGL(k_jac_apply, dim3(nsr_div_up(n, JAC_BLOCK)), dim3(JAC_BLOCK), lds, (hipStream_t)stream, jac, n, desc->n_levels,
                       desc->n_features, dx ? dy : nullptr, dy_stride, dx, d_dy ? g : nullptr, d_dy, d_dy_stride, n_dev);
    NSR_CHECK_LAUNCH("nsr_hashgrid_jac_apply");
    return NSR_OK;
}

extern "C" int nsr_hashgrid_forward_taps(const float *x7, const nsr_half *table, nsr_half *y, uint32_t n,
                                         uint32_t y_stride, int y_level_major, uint32_t level_mask_count,
                                         const NsrGridDesc *desc, const int32_t *n_dev, void *stream)
{
    if (int rc = check_desc(desc, "nsr_hashgrid_forward_taps")) return rc;
    NSR_REQUIRE(y_level_major || y_stride >= desc->n_levels * desc->n_features,
                "nsr_hashgrid_forward_taps: y_stride too small");
    if (n == 0) return NSR_OK;
    NSR_REQUIRE(x7 && table && y, "nsr_hashgrid_forward_taps: NULL pointer");
    const uint32_t lpx = (desc->n_levels + 7) / 8;
    const uint32_t grid = 8u * lpx * nsr_div_up(n, GRID_BLOCK);
    DISPATCH_F(desc->n_features,
               hipLaunchKernelGGL((k_grid_forward_taps<F>), dim3(grid), dim3(GRID_BLOCK), 0, (hipStream_t)stream, x7,
                                  (const __half *)table, (__half *)y, n, y_stride, level_mask_count, lpx, y_level_major,
                                  *desc, n_dev));
    NSR_CHECK_LAUNCH("nsr_hashgrid_forward_taps");
    return NSR_OK;
}

extern "C" int nsr_hashgrid_backward_params(const float *x, const void *dy, int dy_is_f32, uint32_t dy_stride,
                                            float *grad_table, uint32_t n, uint32_t level_mask_count,
                                            float grad_scale, const NsrGridDesc *desc, void *stream)
{
    if (int rc = check_desc(desc, "nsr_hashgrid_backward_params")) return rc;
    if (n == 0) return NSR_OK;
    NSR_REQUIRE(x && dy && grad_table, "nsr_hashgrid_backward_params: NULL pointer");
    const uint32_t lpx = (desc->n_levels + 7) / 8;
    const uint32_t grid = 8u * lpx * nsr_div_up(n, GRID_BLOCK);
    DISPATCH_F(desc->n_features, {
        if (dy_is_f32)
            hipLaunchKernelGGL((k_grid_backward_params<F, true>), dim3(grid), dim3(GRID_BLOCK), 0, (hipStream_t)stream,
                               x, dy, dy_stride, grad_table, n, level_mask_count, lpx, grad_scale, *desc);
        else
            hipLaunchKernelGGL((k_grid_backward_params<F, false>), dim3(grid), dim3(GRID_BLOCK), 0,
                               (hipStream_t)stream, x, dy, dy_stride, grad_table, n, level_mask_count, lpx,
                               grad_scale, *desc);
    });
    NSR_CHECK_LAUNCH("nsr_hashgrid_backward_params");
    return NSR_OK;
}


// workspace (4-byte words): [slabs][level-major dy: L*F*n][counts | bin_start | cursors: n_bins each][items: L*8n]
extern "C" uint64_t nsr_hashgrid_backward_params_workspace_floats(const NsrGridDesc *desc, uint32_t n)
{
    if (!desc || check_desc(desc, "nsr_hashgrid_backward_params_workspace_floats")) return 0;
    OwnerMap om;
    uint64_t slab = 0;
    uint32_t n_bins = 0;
    make_owner_map(desc, &om, &slab, &n_bins);
    return slab + (uint64_t)desc->n_levels * desc->n_features * n + 3ull * n_bins + (uint64_t)desc->n_levels * 8ull * n;
}

// phases: 1 = bin the items (needs only x), 2 = accumulate (needs dy and the bins), 3 = both
static int owner_backward(const float *x, const void *dy, int dy_layout, uint32_t dy_stride, float *grad_table,
                          float *workspace, uint32_t n, uint32_t level_mask_count, float grad_scale, int accumulate,
                          const NsrGridDesc *desc, const int32_t *n_dev, int phases, void *stream,
                          const float *dir = nullptr, const float *dy_first_lm = nullptr,
                          const NsrTableAdam *adam = nullptr, uint32_t taps_nc = 0, float *tap_ws = nullptr,
                          uint16_t *grad_bf16 = nullptr, uint32_t level_begin = 0, uint32_t level_end = 0xffffffffu)
{
    if (int rc = check_desc(desc, "nsr_hashgrid_backward_params_owner")) return rc;
    NSR_REQUIRE(workspace, "nsr_hashgrid_backward_params_owner: workspace is NULL");
    NSR_REQUIRE(n == 0 || x, "nsr_hashgrid_backward_params_owner: NULL pointer");
    NSR_REQUIRE(n < (1u << 28), "nsr_hashgrid_backward_params_owner: at most 2^28 - 1 samples per call");
    const uint32_t F = desc->n_features, L = desc->n_levels;
    hipStream_t st = (hipStream_t)stream;
    OwnerMap om;
    uint64_t slab_floats = 0;
    uint32_t n_bins = 0;
    uint32_t nb = make_owner_map(desc, &om, &slab_floats, &n_bins);
    for (uint32_t l = 0; l < L; ++l)
        NSR_REQUIRE(om.n_slices[l] <= (uint32_t)OWN_MAX_SLICES, "nsr_hashgrid_backward_params_owner: level too large");
    // a launch may cover the run of levels [level_begin, level_end) only (the multi-GPU step exchanges the finest levels'
    // gradient while the coarse ones are still being accumulated)
    if (level_end > L) level_end = L;
    NSR_REQUIRE(level_begin < level_end, "nsr_hashgrid_backward_params_owner: empty level range");
    const bool partial = level_begin > 0 || level_end < L;
    NSR_REQUIRE(!partial || (dy_layout == 2 && !adam && taps_nc == 0 && !dir && (phases & 1) == 0),
                "nsr_hashgrid_backward_params_owner: a level range takes level-major dy, binned items and plain mode");
    const uint32_t row_base = om.level_start[level_begin];
    nb = ((level_end < L ? om.level_start[level_end] : nb / 8u) - row_base) * 8u;
    float *lm = workspace + slab_floats;
    uint32_t *counts = reinterpret_cast<uint32_t *>(lm + (uint64_t)L * F * n);
    uint32_t *bin_start = counts + n_bins, *cursors = bin_start + n_bins, *items = cursors + n_bins;
    // stencil mode (n = 7 taps_nc points, layout [7][taps_nc]): crossing masks | G0 | D in the tap workspace
    NSR_REQUIRE(taps_nc == 0 || (tap_ws && n == 7u * taps_nc && !dir && !adam),
                "nsr_hashgrid_backward_params_owner: the stencil mode takes 7 n_centre points and a tap workspace");
    uint8_t *cross = reinterpret_cast<uint8_t *>(tap_ws);
    float *tap_g0 = tap_ws ? tap_ws + ((uint64_t)L * taps_nc + 15) / 16 * 4 : nullptr;  // (16-byte aligned behind the masks)
    float *tap_dd = tap_ws ? tap_g0 + (uint64_t)L * taps_nc * F : nullptr;
    if (phases & 1) {  // bin the (sample, corner pair) items by owning slice: count, scan, fill
        NSR_REQUIRE(hipMemsetAsync(counts, 0, n_bins * sizeof(uint32_t), st) == hipSuccess,
                    "nsr_hashgrid_backward_params_owner: hipMemsetAsync failed");
        if (n > 0) {
            const uint8_t *cr = taps_nc ? cross : nullptr;
            if (taps_nc)
                hipLaunchKernelGGL(k_tap_cross, dim3(nsr_div_up(taps_nc, 256), L), dim3(256), 0, st, x, taps_nc,
                                   level_mask_count, cross, *desc);
            const dim3 bin_grid(nsr_div_up(n, OWN_BIN_BLOCK * OWN_BIN_SPT), L);
            hipLaunchKernelGGL((k_own_bin<false>), bin_grid, dim3(OWN_BIN_BLOCK), 0, st, x, n, level_mask_count, counts,
                               bin_start, cursors, items, om, *desc, n_dev, cr, taps_nc);
            hipLaunchKernelGGL(k_own_bin_scan, dim3(L), dim3(256), 0, st, counts, bin_start, cursors, om);
            hipLaunchKernelGGL((k_own_bin<true>), bin_grid, dim3(OWN_BIN_BLOCK), 0, st, x, n, level_mask_count, counts,
                               bin_start, cursors, items, om, *desc, n_dev, cr, taps_nc);
            NSR_CHECK_LAUNCH("nsr_hashgrid_backward_params_owner(bin)");
        }
    }
    if (!(phases & 2)) return NSR_OK;
    NSR_REQUIRE((grad_table || adam || grad_bf16) && (n == 0 || dy), "nsr_hashgrid_backward_params_owner: NULL pointer");
    NSR_REQUIRE(!grad_bf16 || (!adam && !accumulate && ((uintptr_t)grad_bf16 & 7) == 0),
                "nsr_hashgrid_backward_params_owner: the bf16 gradient is written once (no accumulate, no fused AdamW) "
                "into an 8-byte aligned buffer");
    OwnerAdam ad;
    memset(&ad, 0, sizeof(ad));
    if (adam) {
        NSR_REQUIRE(adam->params && adam->exp_avg && adam->exp_avg_sq && adam->step && adam->hyper && !accumulate,
                    "nsr_hashgrid_backward_params_owner: fused AdamW needs params / moments / schedule state and "
                    "accumulate == 0");
        NSR_REQUIRE((((uintptr_t)adam->params | (uintptr_t)adam->exp_avg | (uintptr_t)adam->exp_avg_sq) & 15) == 0 &&
                        ((uintptr_t)adam->shadow & 7) == 0 && ((uintptr_t)adam->hyper & 7) == 0,
                    "nsr_hashgrid_backward_params_owner: fused AdamW buffers must be 16-byte aligned (fp16 image: 8)");
        ad.p = adam->params; ad.m = adam->exp_avg; ad.v = adam->exp_avg_sq; ad.shadow = (__half *)adam->shadow;
        ad.step = adam->step; ad.hyper = adam->hyper;
        ad.base_lr = adam->base_lr; ad.b1d = adam->beta1; ad.b2d = adam->beta2; ad.gamma = adam->gamma;
        ad.m0 = adam->milestone0; ad.m1 = adam->milestone1; ad.m2 = adam->milestone2;
        ad.b1 = (float)adam->beta1; ad.b2 = (float)adam->beta2; ad.eps = adam->eps; ad.wd = adam->weight_decay;
    }
    NSR_REQUIRE(dy_layout >= 0 && dy_layout <= 2, "nsr_hashgrid_backward_params_owner: dy_layout must be 0 (half "
                "row-major), 1 (float row-major) or 2 (float level-major)");
    const float *dy_lm = (const float *)dy;
    if (dy_layout != 2 && n > 0) {
        const uint32_t C = L * F;
        const size_t lds = 64 * (C + 1) * sizeof(float);
        DISPATCH_F(F, {
            if (dy_layout == 1)
                hipLaunchKernelGGL((k_dy_to_level_major<F, true>), dim3(nsr_div_up(n, 64)), dim3(256), lds, st, dy,
                                   dy_stride, lm, n, L);
            else
                hipLaunchKernelGGL((k_dy_to_level_major<F, false>), dim3(nsr_div_up(n, 64)), dim3(256), lds, st, dy,
                                   dy_stride, lm, n, L);
        });
        NSR_CHECK_LAUNCH("nsr_hashgrid_backward_params_owner(transpose)");
        dy_lm = lm;
    }
    const size_t lds = OWN_LDS_WORDS * sizeof(unsigned long long);
    DISPATCH_F(F, {
        static bool attr_set = false;  // per instantiation
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void *)k_grid_backward_owner<F, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute((const void *)k_grid_backward_owner<F, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute((const void *)k_grid_backward_owner<F, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_set = true;
        }
        if (taps_nc) {
            NSR_REQUIRE(n_dev == nullptr, "nsr_hashgrid_backward_params_owner: the stencil mode takes a host-side count");
            if (n > 0)
                hipLaunchKernelGGL((k_tap_reduce<F>), dim3(nsr_div_up(taps_nc, 256), L), dim3(256), 0, st, x, dy_lm, cross,
                                   taps_nc, level_mask_count, tap_g0, tap_dd, *desc);
            hipLaunchKernelGGL((k_grid_backward_owner<F, 1>), dim3(nb), dim3(OWN_BLOCK), lds, st, x, dy_lm, items, counts,
                               bin_start, grad_table, workspace, n, level_mask_count, grad_scale, accumulate, om, *desc, dir,
                               dy_first_lm, ad, taps_nc, tap_g0, tap_dd, grad_bf16, row_base);
        } else if (dir) {
            hipLaunchKernelGGL((k_grid_backward_owner<F, 2>), dim3(nb), dim3(OWN_BLOCK), lds, st, x, dy_lm, items, counts,
                               bin_start, grad_table, workspace, n, level_mask_count, grad_scale, accumulate, om, *desc, dir,
                               dy_first_lm, ad, 0u, nullptr, nullptr, grad_bf16, row_base);
        } else
        hipLaunchKernelGGL((k_grid_backward_owner<F, 0>), dim3(nb), dim3(OWN_BLOCK), lds, st, x, dy_lm, items, counts,
                           bin_start, grad_table, workspace, n, level_mask_count, grad_scale, accumulate, om, *desc, dir,
                           dy_first_lm, ad, 0u, nullptr, nullptr, grad_bf16, row_base);
        bool slabs_in_range = false;
        for (uint32_t l = level_begin; l < level_end; ++l) slabs_in_range |= om.n_chunks[l] > 1;
        if (slabs_in_range)
            hipLaunchKernelGGL((k_grid_reduce_slabs<F>), dim3(256, level_end - level_begin), dim3(256), 0, st, workspace,
                               grad_table, accumulate, om, *desc, ad, grad_bf16, level_begin);
    });
    NSR_CHECK_LAUNCH("nsr_hashgrid_backward_params_owner");
    return NSR_OK;
}

extern "C" int nsr_hashgrid_backward_params_owner(const float *x, const void *dy, int dy_layout, uint32_t dy_stride,
                                                  float *grad_table, float *workspace, uint32_t n,
                                                  uint32_t level_mask_count, float grad_scale, int accumulate,
                                                  const NsrGridDesc *desc, const int32_t *n_dev, void *stream)
{
    return owner_backward(x, dy, dy_layout, dy_stride, grad_table, workspace, n, level_mask_count, grad_scale, accumulate,
                          desc, n_dev, 3, stream);
}

extern "C" int nsr_hashgrid_backward_params_owner_bin(const float *x, float *workspace, uint32_t n,
                                                      uint32_t level_mask_count, const NsrGridDesc *desc,
                                                      const int32_t *n_dev, void *stream)
{
    return owner_backward(x, nullptr, 2, 0, nullptr, workspace, n, level_mask_count, 1.f, 0, desc, n_dev, 1, stream);
}

extern "C" int nsr_hashgrid_backward_params_owner_accumulate(const float *x, const void *dy, int dy_layout,
                                                             uint32_t dy_stride, float *grad_table, float *workspace,
                                                             uint32_t n, uint32_t level_mask_count, float grad_scale,
                                                             int accumulate, const NsrGridDesc *desc,
                                                             const int32_t *n_dev, void *stream)
{
    return owner_backward(x, dy, dy_layout, dy_stride, grad_table, workspace, n, level_mask_count, grad_scale, accumulate,
                          desc, n_dev, 2, stream);
}

// ... over the levels [level_begin, level_end) only, the gradient either as fp32 (grad_table) or as bf16 (grad_bf16: the
// transport format of the multi-GPU exchange, nsr/parallel.py); items binned beforehand, dy level-major fp32
extern "C" int nsr_hashgrid_backward_params_owner_accumulate_range(const float *x, const float *dy_level_major,
                                                                   float *grad_table, void *grad_bf16, float *workspace,
                                                                   uint32_t n, uint32_t level_mask_count, float grad_scale,
                                                                   uint32_t level_begin, uint32_t level_end,
                                                                   const NsrGridDesc *desc, const int32_t *n_dev,
                                                                   void *stream)
{
    NSR_REQUIRE((grad_table != nullptr) != (grad_bf16 != nullptr),
                "nsr_hashgrid_backward_params_owner_accumulate_range: exactly one of grad_table / grad_bf16");
    return owner_backward(x, dy_level_major, 2, 0, grad_table, workspace, n, level_mask_count, grad_scale, 0, desc, n_dev, 2,
                          stream, nullptr, nullptr, nullptr, 0, nullptr, (uint16_t *)grad_bf16, level_begin, level_end);
}

// ... with AdamW applied to the table by the workgroups that own the slices (see OwnerAdam): no gradient is written
extern "C" int nsr_hashgrid_backward_params_owner_accumulate_adam(const float *x, const void *dy, int dy_layout,
                                                                  uint32_t dy_stride, float *workspace, uint32_t n,
                                                                  uint32_t level_mask_count, float grad_scale,
                                                                  const NsrGridDesc *desc, const int32_t *n_dev,
                                                                  const NsrTableAdam *adam, void *stream)
{
    NSR_REQUIRE(adam, "nsr_hashgrid_backward_params_owner_accumulate_adam: adam is NULL");
    return owner_backward(x, dy, dy_layout, dy_stride, nullptr, workspace, n, level_mask_count, grad_scale, 0, desc, n_dev,
                          2, stream, nullptr, nullptr, adam);
}

// ---- stencil mode: the 7 n_centre points of a finite-difference step (positions [7][n_centre][3]: sample, then the six
// +-eps taps; dy level-major [L][7 n_centre][F]).  Taps that stay in their sample's cell are folded into the sample's items
// (see k_tap_cross / k_tap_reduce); the result equals the plain call over all 7 n_centre points up to fp32 rounding.
extern "C" uint64_t nsr_hashgrid_backward_params_taps_workspace_floats(const NsrGridDesc *desc, uint32_t n_centre)
{
    if (!desc || check_desc(desc, "nsr_hashgrid_backward_params_taps_workspace_floats")) return 0;
    const uint64_t L = desc->n_levels, F = desc->n_features;
    return (L * n_centre + 15) / 16 * 4 + L * n_centre * F * 4;
}

extern "C" int nsr_hashgrid_backward_params_owner_bin_taps(const float *x7, float *workspace, float *tap_workspace,
                                                           uint32_t n_centre, uint32_t level_mask_count,
                                                           const NsrGridDesc *desc, void *stream)
{
    NSR_REQUIRE(n_centre > 0 && tap_workspace, "nsr_hashgrid_backward_params_owner_bin_taps: empty input / NULL workspace");
    return owner_backward(x7, nullptr, 2, 0, nullptr, workspace, 7u * n_centre, level_mask_count, 1.f, 0, desc, nullptr, 1,
                          stream, nullptr, nullptr, nullptr, n_centre, tap_workspace);
}

extern "C" int nsr_hashgrid_backward_params_owner_accumulate_taps(const float *x7, const float *dy_level_major,
                                                                  float *grad_table, float *workspace,
                                                                  float *tap_workspace, uint32_t n_centre,
                                                                  uint32_t level_mask_count, int accumulate,
                                                                  const NsrGridDesc *desc, void *stream)
{
    NSR_REQUIRE(n_centre > 0 && tap_workspace, "nsr_hashgrid_backward_params_owner_accumulate_taps: empty input / NULL workspace");
    return owner_backward(x7, dy_level_major, 2, 0, grad_table, workspace, 7u * n_centre, level_mask_count, 1.f, accumulate,
                          desc, nullptr, 2, stream, nullptr, nullptr, nullptr, n_centre, tap_workspace);
}

// first-order table gradient (dy_first, level-major fp32) and the second-order one of the analytic normal (dy row-major
// fp32 with `g` = dL/d(dx)) in ONE binning + accumulation pass
extern "C" int nsr_hashgrid_backward_params_owner_with_second_order(const float *x, const float *dy_first_lm,
                                                                    const float *dy, uint32_t dy_stride, const float *g,
                                                                    float *grad_table, float *workspace, uint32_t n,
                                                                    uint32_t level_mask_count, int accumulate,
                                                                    int binned, const NsrGridDesc *desc, void *stream)
{
    NSR_REQUIRE(n == 0 || (dy_first_lm && dy && g), "nsr_hashgrid_backward_params_owner_with_second_order: NULL pointer");
    // binned != 0: the items of these positions are already in `workspace` (nsr_hashgrid_backward_params_owner_bin, e.g.
    // queued on a helper stream right after the positions were formed)
    return owner_backward(x, dy, 1, dy_stride, grad_table, workspace, n, level_mask_count, 1.f, accumulate, desc, nullptr,
                          binned ? 2 : 3, stream, g, dy_first_lm);
}

extern "C" int nsr_hashgrid_backward_input(const float *x, const nsr_half *table, const void *dy, int dy_is_f32,
                                           uint32_t dy_stride, float *dx, uint32_t n, uint32_t level_mask_count,
                                           const NsrGridDesc *desc, void *stream)
{
    if (int rc = check_desc(desc, "nsr_hashgrid_backward_input")) return rc;
    if (n == 0) return NSR_OK;
    NSR_REQUIRE(x && table && dy && dx, "nsr_hashgrid_backward_input: NULL pointer");
    const uint32_t grid = nsr_div_up(n, GRID_BLOCK);
    DISPATCH_F(desc->n_features, {
        if (dy_is_f32)
            hipLaunchKernelGGL((k_grid_backward_input<F, true>), dim3(grid), dim3(GRID_BLOCK), 0, (hipStream_t)stream,
                               x, (const __half *)table, dy, dy_stride, dx, n, level_mask_count, *desc);
        else
            hipLaunchKernelGGL((k_grid_backward_input<F, false>), dim3(grid), dim3(GRID_BLOCK), 0, (hipStream_t)stream,
                               x, (const __half *)table, dy, dy_stride, dx, n, level_mask_count, *desc);
    });
    NSR_CHECK_LAUNCH("nsr_hashgrid_backward_input");
    return NSR_OK;
}

extern "C" int nsr_hashgrid_backward_backward_input(const float *x, const nsr_half *table, const void *dy,
                                                    int dy_is_f32, uint32_t dy_stride, const float *g, float *d_dy,
                                                    uint32_t d_dy_stride, float *grad_table, float *dx2, uint32_t n,
                                                    uint32_t level_mask_count, const NsrGridDesc *desc, void *stream)
{
    if (int rc = check_desc(desc, "nsr_hashgrid_backward_backward_input")) return rc;
    if (n == 0) return NSR_OK;
    NSR_REQUIRE(x && table && dy && g, "nsr_hashgrid_backward_backward_input: NULL pointer");
    const uint32_t grid = nsr_div_up(n, GRID_BLOCK);
    DISPATCH_F(desc->n_features, {
        if (dy_is_f32)
            hipLaunchKernelGGL((k_grid_bwd_bwd_input<F, true>), dim3(grid), dim3(GRID_BLOCK), 0, (hipStream_t)stream, x,
                               (const __half *)table, dy, dy_stride, g, d_dy, d_dy_stride, grad_table, dx2, n,
                               level_mask_count, *desc);
        else
            hipLaunchKernelGGL((k_grid_bwd_bwd_input<F, false>), dim3(grid), dim3(GRID_BLOCK), 0, (hipStream_t)stream,
                               x, (const __half *)table, dy, dy_stride, g, d_dy, d_dy_stride, grad_table, dx2, n,
                               level_mask_count, *desc);
    });
    NSR_CHECK_LAUNCH("nsr_hashgrid_backward_backward_input");
    return NSR_OK;
}

// The same with the table gradient accumulated by the binned owner-computes path instead of 2 * 8 * L global float
// atomics per sample (measured on the NeuS step, 2.3e6 samples: 28.9 ms of its 63.6 ms were this kernel's atomics).
// workspace: nsr_hashgrid_backward_params_workspace_floats(desc, n) floats.  grad_table is ACCUMULATED into.
extern "C" int nsr_hashgrid_backward_backward_input_ws(const float *x, const nsr_half *table, const void *dy,
                                                       int dy_is_f32, uint32_t dy_stride, const float *g, float *d_dy,
                                                       uint32_t d_dy_stride, float *grad_table, float *dx2,
                                                       float *workspace, uint32_t n, uint32_t level_mask_count,
                                                       const NsrGridDesc *desc, void *stream)
{
    if (!grad_table || !workspace)
        return nsr_hashgrid_backward_backward_input(x, table, dy, dy_is_f32, dy_stride, g, d_dy, d_dy_stride, grad_table,
                                                    dx2, n, level_mask_count, desc, stream);
    if (d_dy || dx2)
        if (int rc = nsr_hashgrid_backward_backward_input(x, table, dy, dy_is_f32, dy_stride, g, d_dy, d_dy_stride,
                                                          nullptr, dx2, n, level_mask_count, desc, stream))
            return rc;
    if (n == 0) return NSR_OK;
    return owner_backward(x, dy, dy_is_f32 ? 1 : 0, dy_stride, grad_table, workspace, n, level_mask_count, 1.f, 1, desc,
                          nullptr, 3, stream, g);
}
