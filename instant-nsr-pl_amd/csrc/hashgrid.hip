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
#include <stdlib.h>

#include "nsr_common.h"
#include "hashgrid_geom.h"

namespace {

constexpr int GRID_BLOCK = 256;

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

// Where row i, level `level` of an encoding with `rows` rows goes.  layout 0: row-major [rows, y_stride] (what the tcnn API
// returns; a wave's stores are 64 F-half pieces 2 * y_stride bytes apart -- each leaves the L2 as its own masked sector:
// measured 160-187 MB of fabric writes for a 17-19 MB output).  1: level-major [L][rows][F] (a wave stores 64 x F
// consecutive halfs; what the NeRF step and the table backward use).  2: tile-major [rows / 16][L][16][F]: the 16 rows of
// an MFMA tile are ONE contiguous L * 32 F bytes (as in row-major), and inside it each level's 16 x F halfs are
// contiguous -- the encode still stores 64-B runs, the fp32 MLP kernels (csrc/vmlp.hip, lane = (row in tile, k mod 4))
// read 2-3 such runs per load instruction instead of 16 rows.
__device__ __forceinline__ __half *enc_at(__half *y, int layout, uint64_t i, uint32_t level, uint64_t rows, uint32_t y_stride,
                                          uint32_t L, uint32_t F)
{
    if (layout == 1) return y + ((uint64_t)level * rows + i) * F;
    if (layout == 2) return y + (((i >> 4) * L + level) * 16 + (i & 15)) * F;
    return y + i * y_stride + level * F;
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
    __half *yo = enc_at(y, level_major, i, level, n, y_stride, d.n_levels, F);
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
        store_enc<F>(enc_at(y, level_major, i, level, n, y_stride, d.n_levels, F), acc);
    }
}

// blocks b with (b & 7) = xcd, q = b >> 3: both levels {xcd, xcd + 8} of sample block q (levels < level_begin skipped)
template <int F>
__global__ void __launch_bounds__(GRID_BLOCK)
k_grid_forward_pair(const float *__restrict__ x, const __half *__restrict__ table, __half *__restrict__ y, uint32_t n,
                    uint32_t y_stride, uint32_t mask_count, uint32_t level_begin, int level_major, const NsrGridDesc d,
                    const int32_t *__restrict__ n_dev)
{
    const uint32_t xcd = blockIdx.x & 7u;
    const uint32_t blk = blockIdx.x >> 3;
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
        store_enc<F>(enc_at(y, level_major, i, level, n, y_stride, d.n_levels, F), acc);
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

// the corners (cx, cy, cz) and (cx + 1, cy, cz).  F == 2, hashed level, even cx: they are the two halves of one aligned 8-byte
// pair (hash(cx | 1, ..) = hash(cx, ..) ^ 1, see k_grid_forward) -- one gather instead of two
template <int F>
__device__ __forceinline__ void load_x_pair(const __half *__restrict__ table, const LevelGeom &g, uint32_t cx, uint32_t cy,
                                            uint32_t cz, bool paired, float (&v0)[F], float (&v1)[F])
{
    if constexpr (F == 2) {
        if (paired) {
            const uint32_t e0 = corner_index(g, cx, cy, cz);
            const uint2 raw = *reinterpret_cast<const uint2 *>(table + (uint64_t)(g.offset + (e0 & ~1u)) * 2);
            const __half2 lo = *reinterpret_cast<const __half2 *>(&raw.x), hi = *reinterpret_cast<const __half2 *>(&raw.y);
            const __half2 a = (e0 & 1u) ? hi : lo, b = (e0 & 1u) ? lo : hi;  // entry e0, entry e0 ^ 1
            v0[0] = __low2float(a); v0[1] = __high2float(a);
            v1[0] = __low2float(b); v1[1] = __high2float(b);
            return;
        }
    }
    load_feat<F>(table, g.offset + corner_index(g, cx, cy, cz), v0);
    load_feat<F>(table, g.offset + corner_index(g, cx + 1u, cy, cz), v1);
}

template <int F>
__global__ void __launch_bounds__(GRID_BLOCK)
k_grid_forward_taps(const float *__restrict__ x7, const __half *__restrict__ table, __half *__restrict__ y, uint32_t n,
                    uint32_t y_stride, uint32_t mask_count, uint32_t lpx, int level_major, const NsrGridDesc d,
                    const int32_t *__restrict__ n_dev,
                    uint8_t *__restrict__ cross /* [L][n] or NULL: bit t = tap t + 1 left the sample's cell on this level --
                                                   what the stencil mode of the table backward bins by (k_tap_cross) */)
{
    // row pointer of point p (0 .. 7n-1): row-major [7n][y_stride] or level-major [L][7n][F]
    const uint64_t n7 = 7ull * n;
#define TAP_ROW(p) enc_at(y, level_major, (p), level, n7, y_stride, d.n_levels, F)
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
    // the forward is bound by the L2 request rate: x-neighbours of an even cell of a hashed level come as one 8-byte gather
    // (the y / z taps stay in the sample's x column, so their far faces pair up the same way)
    const bool paired = F == 2 && !g.dense && !(cb.c[0] & 1u);
    float vb[8][F];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        load_x_pair<F>(table, g, cb.c[0], cb.c[1] + (j & 1), cb.c[2] + (j >> 1), paired, vb[2 * j], vb[2 * j + 1]);
    blend8<F>(cb, vb, TAP_ROW(i));
    uint32_t cross_mask = 0u;
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const int a = t >> 1;  // the axis this tap moved along
        float xt[3] = {xb[0], xb[1], xb[2]};
        xt[a] = x7[((uint64_t)(t + 1) * n + i) * 3 + a];
        const Cell ct = locate(g, xt[0], xt[1], xt[2]);
        const int dc = (int)ct.c[a] - (int)cb.c[a];
        cross_mask |= dc != 0 ? 1u << t : 0u;
        float vt[8][F];
        if (dc == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int f = 0; f < F; ++f) vt[k][f] = vb[k][f];
        } else if (dc == 1 || dc == -1) {
            // corner k of the tap's cell with bit a == (dc < 0) is corner k ^ (1 << a) of the sample's cell (shared face)
            const int shared_bit = dc < 0 ? 1 : 0;
            if (a == 0) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if ((k & 1) == shared_bit) {
#pragma unroll
                        for (int f = 0; f < F; ++f) vt[k][f] = vb[k ^ 1][f];
                    } else {
                        load_feat<F>(table, g.offset + corner_index(g, ct.c[0] + (k & 1), ct.c[1] + ((k >> 1) & 1),
                                                                    ct.c[2] + ((k >> 2) & 1)), vt[k]);
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {  // corners 2j, 2j + 1: the x pair at (y bit, z bit) = (j & 1, j >> 1)
                    if ((((2 * j) >> a) & 1) == shared_bit) {
#pragma unroll
                        for (int f = 0; f < F; ++f) {
                            vt[2 * j][f] = vb[(2 * j) ^ (1 << a)][f];
                            vt[2 * j + 1][f] = vb[(2 * j + 1) ^ (1 << a)][f];
                        }
                    } else {
                        load_x_pair<F>(table, g, ct.c[0], ct.c[1] + (j & 1), ct.c[2] + (j >> 1), paired, vt[2 * j], vt[2 * j + 1]);
                    }
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
    if (cross) cross[(uint64_t)level * n + i] = (uint8_t)cross_mask;
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
            uint32_t d_dy_stride, const int32_t *__restrict__ n_dev,
            float *__restrict__ dy_lm_out /* with dx: also leave dy level-major [L][n][F] (the tile is in LDS anyway) */)
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
        if (dy_lm_out) {  // what the table backward's second-order term reads: saves it a transposing pass over dy
            const uint32_t per = rows * F;
            for (uint32_t k = threadIdx.x; k < L * per; k += JAC_BLOCK) {
                const uint32_t l = k / per, rem = k % per, r = rem / F, f = rem % F;
                dy_lm_out[((uint64_t)l * n + i0 + r) * F + f] = tile[r * ld + l * F + f];
            }
        }
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
    return nsr_hashgrid_jac_apply_ex(jac, n, desc, dy, dy_stride, dx, g, d_dy, d_dy_stride, nullptr, n_dev, stream);
}

extern "C" int nsr_hashgrid_jac_apply_ex(const float *jac, uint32_t n, const NsrGridDesc *desc, const float *dy,
                                         uint32_t dy_stride, float *dx, const float *g, float *d_dy, uint32_t d_dy_stride,
                                         float *dy_level_major_out, const int32_t *n_dev, void *stream)
{
    if (int rc = check_desc(desc, "nsr_hashgrid_jac_apply")) return rc;
    NSR_REQUIRE(!dy_level_major_out || (dx && dy), "nsr_hashgrid_jac_apply: the level-major copy is of dy (needs dx, dy)");
    if (n == 0) return NSR_OK;
    NSR_REQUIRE(jac && ((dx && dy) || (d_dy && g)), "nsr_hashgrid_jac_apply: NULL pointer");
    const size_t lds = (size_t)JAC_BLOCK * (desc->n_levels * desc->n_features + 1) * sizeof(float);
    hipLaunchKernelGGL(k_jac_apply, dim3(nsr_div_up(n, JAC_BLOCK)), dim3(JAC_BLOCK), lds, (hipStream_t)stream, jac, n, desc->n_levels,
                       desc->n_features, dx ? dy : nullptr, dy_stride, dx, d_dy ? g : nullptr, d_dy, d_dy_stride, n_dev,
                       dy_level_major_out);
    NSR_CHECK_LAUNCH("nsr_hashgrid_jac_apply");
    return NSR_OK;
}

static int forward_taps(const float *x7, const nsr_half *table, nsr_half *y, uint32_t n, uint32_t y_stride, int y_level_major,
                        uint32_t level_mask_count, const NsrGridDesc *desc, const int32_t *n_dev, uint8_t *cross, void *stream)
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
                                  *desc, n_dev, cross));
    NSR_CHECK_LAUNCH("nsr_hashgrid_forward_taps");
    return NSR_OK;
}

extern "C" int nsr_hashgrid_forward_taps(const float *x7, const nsr_half *table, nsr_half *y, uint32_t n,
                                         uint32_t y_stride, int y_level_major, uint32_t level_mask_count,
                                         const NsrGridDesc *desc, const int32_t *n_dev, void *stream)
{
    return forward_taps(x7, table, y, n, y_stride, y_level_major, level_mask_count, desc, n_dev, nullptr, stream);
}

// ... and leave the per-(level, sample) crossing masks of the six taps at the head of the table backward's tap workspace
// (nsr_hashgrid_backward_params_taps_workspace_floats for the same n): the lane already holds both cells, and
// nsr_hashgrid_backward_params_owner_bin_taps_masked then needs no pass of its own over the 7 n positions x L levels
extern "C" int nsr_hashgrid_forward_taps_masks(const float *x7, const nsr_half *table, nsr_half *y, uint32_t n,
                                               uint32_t y_stride, int y_level_major, uint32_t level_mask_count,
                                               const NsrGridDesc *desc, float *tap_workspace, void *stream)
{
    NSR_REQUIRE(tap_workspace, "nsr_hashgrid_forward_taps_masks: NULL tap workspace");
    return forward_taps(x7, table, y, n, y_stride, y_level_major, level_mask_count, desc, nullptr,
                        reinterpret_cast<uint8_t *>(tap_workspace), stream);
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


// ---- the owner-computes table backward, in two configurations (hashgrid_owner.inc) -------------------------------------
// Measured (tools/table_backward_variants.py, tools/neus_step_bench.py, round 3), accumulate / accumulate + AdamW at 9.6e4
// ray-coherent samples: 2^13-entry slices x 1024 threads (one workgroup per CU) 87 / 122 us, 2^12 x 512 77 / 111 us,
// 2^11 x 256 72 / 105 us -- several small workgroups per CU overlap one's item phase (LDS atomics, latency bound) with
// another's write-out / AdamW phase (HBM streaming).  With ~1e6 points per launch (the NeuS steps at 4096 rays, the 7 N
// points of a finite-difference step) the large slices win again: C3 5.62 vs 6.04 ms, C5 15.3 vs 15.9 ms per step -- fewer,
// longer item lists per workgroup amortise the per-workgroup LDS clear / write-out.  So both are compiled and a launch
// picks by its point count (the binning and the accumulation of one gradient see the same count).
//
// Run-time knobs of the decomposition (nsr_hashgrid_owner_tune; A/B switches of tools/table_backward_variants.py).  The
// gradient of the hashed levels is the same bits under every setting (integer sums); on the dense levels the chunk slabs and
// the row merge add in fp32, so placement / chunking / merge settings change their association (agreement to ~1e-6):
struct OwnTune {
    int placement;         // 0: units dealt over the XCDs, 1: contiguous cost-balanced ranges, 2: levels striped over XCD pairs, 3: striped lists claimed at run time, 4: 3 for the large configuration / 2 for the small one (default)
    float cost_adam;       // weight of a unit's write-out share in the balance
    float cost_items;      // weight of a unit's item share
    uint32_t dense_epb_log2;  // entries per slice of a dense level (log2)
    uint32_t dense_wgs;       // workgroups a dense level is cut into at least (slices x item chunks)
    uint32_t merge_dense;     // 2: every dense level merges runs of same-entry lanes in fp32 registers, 1: the chunked ones (default), 0: none
};
static OwnTune g_own_tune = {4, 1.0f, 3.0f, 11u, 64u, 1u};

// The claimed placement's cursors: five words per launch, zero when the launch starts and cleared again by the workgroup that
// takes its last unit.  Launches in flight at the same time (two tables on two streams, a step queued behind the previous
// one) must not share them, so every launch takes the next of 256 slots of a per-device pool -- a slot comes round again
// 256 owner launches later, long after its launch has drained.
static uint32_t *owner_claim_slot()
{
    constexpr int SLOTS = 256, WORDS = 8, MAX_DEV = 16;
    static uint32_t *pool[MAX_DEV] = {nullptr};
    static uint32_t next[MAX_DEV] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return nullptr;
    if (!pool[dev]) {
        uint32_t *p = nullptr;
        if (hipMalloc(&p, SLOTS * WORDS * sizeof(uint32_t)) != hipSuccess) return nullptr;
        if (hipMemset(p, 0, SLOTS * WORDS * sizeof(uint32_t)) != hipSuccess) { (void)hipFree(p); return nullptr; }
        pool[dev] = p;
    }
    return pool[dev] + WORDS * (next[dev]++ % SLOTS);
}

#ifndef NSR_OWN_SMALL_LOG2
#define NSR_OWN_SMALL_LOG2 11
#endif
#ifndef NSR_OWN_SMALL_BLOCK
#define NSR_OWN_SMALL_BLOCK 256
#endif
namespace own_small {
#define NSR_OWN_BLOCK NSR_OWN_SMALL_BLOCK
#define NSR_OWN_LOG2 NSR_OWN_SMALL_LOG2
#include "hashgrid_owner.inc"
#undef NSR_OWN_BLOCK
#undef NSR_OWN_LOG2
}  // namespace own_small
namespace own_large {
#define NSR_OWN_BLOCK 1024
#define NSR_OWN_LOG2 13
#include "hashgrid_owner.inc"
#undef NSR_OWN_BLOCK
#undef NSR_OWN_LOG2
}  // namespace own_large

#ifndef NSR_OWN_LARGE_FROM
#define NSR_OWN_LARGE_FROM 400000u
#endif
static uint32_t g_own_large_from = NSR_OWN_LARGE_FROM;
// g_own_second_order: the launch (binning or accumulation) belongs to a first + second-order pass -- its items cost about
// twice a plain one's (two more gathers, the directional-derivative weights), so the large configuration pays from half
// the point count (measured at the NeuS operating point, 2.6e5 samples: C3 step 1.457 -> 1.415 ms)
static thread_local bool g_own_second_order = false;
static bool own_use_large(uint32_t n) { return (g_own_second_order ? 2ull * n : (uint64_t)n) > g_own_large_from; }
struct SecondOrderScope {
    bool old;
    explicit SecondOrderScope(bool on) : old(g_own_second_order) { g_own_second_order = on; }
    ~SecondOrderScope() { g_own_second_order = old; }
};

// launches of more than `n_points` points use the large-slice configuration (0: always, UINT32_MAX: never); returns the
// previous threshold.  The binning and the accumulation of one gradient must see the same setting.
extern "C" uint32_t nsr_hashgrid_owner_large_from(uint32_t n_points)
{
    const uint32_t old = g_own_large_from;
    g_own_large_from = n_points;
    return old;
}

// key 0: placement (0 dealt / 1 listed / 2 striped / 3 claimed / 4 by configuration), 1: cost_adam, 2: cost_items, 3: dense_epb_log2, 4: dense_wgs, 5: merge_dense
extern "C" float nsr_hashgrid_owner_tune(int key, float value)
{
    float old = 0.f;
    switch (key) {
    case 0: old = (float)g_own_tune.placement; g_own_tune.placement = value < 0.5f ? 0 : (value < 1.5f ? 1 : (value < 2.5f ? 2 : (value < 3.5f ? 3 : 4))); break;
    case 1: old = g_own_tune.cost_adam; g_own_tune.cost_adam = value; break;
    case 2: old = g_own_tune.cost_items; g_own_tune.cost_items = value; break;
    case 3: old = (float)g_own_tune.dense_epb_log2; g_own_tune.dense_epb_log2 = (uint32_t)value; break;
    case 4: old = (float)g_own_tune.dense_wgs; g_own_tune.dense_wgs = (uint32_t)value; break;
    case 5: old = (float)g_own_tune.merge_dense; g_own_tune.merge_dense = (uint32_t)value; break;
    default: break;
    }
    return old;
}

// the owner's part of the workspace (either configuration), rounded to 16 bytes: where the dense-level accumulators start
static uint64_t owner_workspace_floats(const NsrGridDesc *desc, uint32_t n)
{
    const uint64_t a = own_small::workspace_floats(desc, n), b = own_large::workspace_floats(desc, n);
    return ((a > b ? a : b) + 3ull) & ~3ull;  // (the configuration is picked per launch: room for either)
}

extern "C" uint64_t nsr_hashgrid_backward_params_workspace_floats(const NsrGridDesc *desc, uint32_t n)
{
    if (!desc || check_desc(desc, "nsr_hashgrid_backward_params_workspace_floats")) return 0;
    return owner_workspace_floats(desc, n);
}


template <typename... A>
static int owner_backward(const float *x, const void *dy, int dy_layout, uint32_t dy_stride, float *grad_table,
                          float *workspace, uint32_t n, A... rest)
{
    return own_use_large(n) ? own_large::owner_backward(x, dy, dy_layout, dy_stride, grad_table, workspace, n, rest...)
                            : own_small::owner_backward(x, dy, dy_layout, dy_stride, grad_table, workspace, n, rest...);
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

// ... for items that a ..._with_second_order accumulation (binned != 0) will consume: the slice configuration of a launch is
// picked from the point count AND the kind of pass, and binning and accumulation must agree on it
extern "C" int nsr_hashgrid_backward_params_owner_bin_second_order(const float *x, float *workspace, uint32_t n,
                                                                   uint32_t level_mask_count, const NsrGridDesc *desc,
                                                                   const int32_t *n_dev, void *stream)
{
    SecondOrderScope scope(true);
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

// ... when nsr_hashgrid_forward_taps_masks already left the crossing masks in tap_workspace
extern "C" int nsr_hashgrid_backward_params_owner_bin_taps_masked(const float *x7, float *workspace, float *tap_workspace,
                                                                  uint32_t n_centre, uint32_t level_mask_count,
                                                                  const NsrGridDesc *desc, void *stream)
{
    NSR_REQUIRE(n_centre > 0 && tap_workspace,
                "nsr_hashgrid_backward_params_owner_bin_taps_masked: empty input / NULL workspace");
    return owner_backward(x7, nullptr, 2, 0, nullptr, workspace, 7u * n_centre, level_mask_count, 1.f, 0, desc, nullptr, 5,
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

// ... with AdamW applied to the table in the write-out (NsrTableAdam): no gradient is stored
extern "C" int nsr_hashgrid_backward_params_owner_accumulate_taps_adam(const float *x7, const float *dy_level_major,
                                                                       float *workspace, float *tap_workspace,
                                                                       uint32_t n_centre, uint32_t level_mask_count,
                                                                       const NsrGridDesc *desc, const NsrTableAdam *adam,
                                                                       void *stream)
{
    NSR_REQUIRE(n_centre > 0 && tap_workspace && adam,
                "nsr_hashgrid_backward_params_owner_accumulate_taps_adam: empty input / NULL workspace / NULL adam");
    return owner_backward(x7, dy_level_major, 2, 0, nullptr, workspace, 7u * n_centre, level_mask_count, 1.f, 0, desc, nullptr,
                          2, stream, nullptr, nullptr, adam, n_centre, tap_workspace);
}

// ... or written as bf16 into the multi-GPU exchange's send buffer (nsr/parallel.py): no fp32 gradient, no cast kernel
extern "C" int nsr_hashgrid_backward_params_owner_accumulate_taps_bf16(const float *x7, const float *dy_level_major,
                                                                       uint16_t *grad_bf16, float *workspace,
                                                                       float *tap_workspace, uint32_t n_centre,
                                                                       uint32_t level_mask_count, const NsrGridDesc *desc,
                                                                       void *stream)
{
    NSR_REQUIRE(n_centre > 0 && tap_workspace && grad_bf16,
                "nsr_hashgrid_backward_params_owner_accumulate_taps_bf16: empty input / NULL workspace / NULL output");
    return owner_backward(x7, dy_level_major, 2, 0, nullptr, workspace, 7u * n_centre, level_mask_count, 1.f, 0, desc, nullptr,
                          2, stream, nullptr, nullptr, nullptr, n_centre, tap_workspace, grad_bf16);
}

// first-order table gradient (dy_first, level-major fp32) and the second-order one of the analytic normal (dy row-major
// fp32 with `g` = dL/d(dx)) in ONE binning + accumulation pass
extern "C" int nsr_hashgrid_backward_params_owner_with_second_order(const float *x, const float *dy_first_lm,
                                                                    const float *dy, uint32_t dy_stride, const float *g,
                                                                    float *grad_table, float *workspace, uint32_t n,
                                                                    uint32_t level_mask_count, int accumulate,
                                                                    int binned, const NsrGridDesc *desc, void *stream)
{
    SecondOrderScope scope(true);
    NSR_REQUIRE(n == 0 || (dy_first_lm && dy && g), "nsr_hashgrid_backward_params_owner_with_second_order: NULL pointer");
    // binned != 0: the items of these positions are already in `workspace` (nsr_hashgrid_backward_params_owner_bin, e.g.
    // queued on a helper stream right after the positions were formed)
    return owner_backward(x, dy, dy_stride ? 1 : 2, dy_stride, grad_table, workspace, n, level_mask_count, 1.f, accumulate, desc, nullptr,
                          binned ? 2 : 3, stream, g, dy_first_lm);
}

extern "C" int nsr_hashgrid_backward_params_owner_with_second_order_adam(const float *x, const float *dy_first_lm,
                                                                         const float *dy, uint32_t dy_stride,
                                                                         const float *g, float *workspace, uint32_t n,
                                                                         uint32_t level_mask_count, int binned,
                                                                         const NsrGridDesc *desc, const NsrTableAdam *adam,
                                                                         void *stream)
{
    SecondOrderScope scope(true);
    NSR_REQUIRE(adam && (n == 0 || (dy_first_lm && dy && g)),
                "nsr_hashgrid_backward_params_owner_with_second_order_adam: NULL pointer");
    return owner_backward(x, dy, dy_stride ? 1 : 2, dy_stride, nullptr, workspace, n, level_mask_count, 1.f, 0, desc, nullptr, binned ? 2 : 3,
                          stream, g, dy_first_lm, adam);
}

extern "C" int nsr_hashgrid_backward_params_owner_with_second_order_bf16(const float *x, const float *dy_first_lm,
                                                                         const float *dy, uint32_t dy_stride,
                                                                         const float *g, uint16_t *grad_bf16,
                                                                         float *workspace, uint32_t n,
                                                                         uint32_t level_mask_count, int binned,
                                                                         const NsrGridDesc *desc, void *stream)
{
    SecondOrderScope scope(true);
    NSR_REQUIRE(grad_bf16 && (n == 0 || (dy_first_lm && dy && g)),
                "nsr_hashgrid_backward_params_owner_with_second_order_bf16: NULL pointer");
    return owner_backward(x, dy, dy_stride ? 1 : 2, dy_stride, nullptr, workspace, n, level_mask_count, 1.f, 0, desc, nullptr, binned ? 2 : 3,
                          stream, g, dy_first_lm, nullptr, 0u, nullptr, grad_bf16);
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

#ifdef NSR_OWN_TIMING
// debug build only (tools/owner_phases.sh): read + clear the per-level phase ticks of the 2^13 x 1,024 configuration
extern "C" int nsr_debug_owner_timing(unsigned long long *out /* [NSR_MAX_LEVELS][8] host */, int large)
{
    const void *sym = large ? (const void *)&own_large::g_own_timing : (const void *)&own_small::g_own_timing;
    if (hipMemcpyFromSymbol(out, sym, sizeof(unsigned long long) * NSR_MAX_LEVELS * 8) != hipSuccess) return NSR_ERR_LAUNCH;
    static unsigned long long zeros[NSR_MAX_LEVELS * 8] = {0};
    if (hipMemcpyToSymbol(sym, zeros, sizeof(zeros)) != hipSuccess) return NSR_ERR_LAUNCH;
    return NSR_OK;
}
#endif
