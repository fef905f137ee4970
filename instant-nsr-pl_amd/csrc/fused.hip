// Fused glue of the NeRF training step for gfx950: the ~100 small elementwise / indexing / autograd kernels that
// the reference runs between its tcnn and nerfacc calls (models/nerf.py:61-127, models/geometry.py:122-130,
// models/texture.py:23-30, systems/nerf.py:33-99) collapsed into a handful of launches.
//
//   sample_positions_unit : o[r] + d[r]*(t0+t1)/2 -> contract_to_unisphere           (nerf.py:66-69,95-99; geometry.py:17-29)
//   visibility_prefix     : sigma_fn -> alpha -> transmittance -> T >= eps, per ray  (nerfacc render_visibility inside
//                           ray_marching; T is non-increasing so the kept samples of a ray are a PREFIX -> counts only)
//   copy_ray_prefixes     : order-preserving compaction of those prefixes
//   texture_input         : [feature | SH4((d+1)/2)] as the fp16 MLP input              (texture.py:24-26)
//   composite_forward     : trunc_exp density -> weights -> opacity/depth/rgb + background blend  (nerf.py:105-109)
//   smooth_l1_valid       : F.smooth_l1_loss(comp_rgb[valid], rgb[valid]) and its gradient        (systems/nerf.py:97)
//   composite_backward    : d comp_rgb -> d rgb, d density-logit (transmittance scan backward + trunc_exp backward)
//   gather_train_rays     : pixel gather + get_rays + normalise + background blend     (systems/nerf.py:38-79)
//
// One wavefront per ray for everything segmented (shuffle scans, no LDS, deterministic order).
#include <string.h>
#include "nsr_common.h"

namespace {

constexpr int EW_BLOCK = 256;
constexpr int R_BLOCK = 256;
constexpr int RAYS_PER_BLOCK = R_BLOCK / NSR_WAVE;

__global__ void __launch_bounds__(EW_BLOCK)
k_sample_positions_unit(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                        const int64_t *__restrict__ ray_indices, const float *__restrict__ t0,
                        const float *__restrict__ t1, float radius, int type, float *__restrict__ x01,
                        float *__restrict__ dirs, uint32_t n, const int32_t *__restrict__ n_dev)
{
    const uint32_t i = blockIdx.x * EW_BLOCK + threadIdx.x;
    if (i >= live_count(n, n_dev)) return;
    const int64_t r = ray_indices[i];
    const float tm = (t0[i] + t1[i]) / 2.f;
    const float den = radius - (-radius);
    float v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float d = rays_d[3 * r + k];
        const float p = __fadd_rn(rays_o[3 * r + k], __fmul_rn(d, tm));
        v[k] = (p - (-radius)) / den;
        if (dirs) dirs[3ull * i + k] = d;
    }
    if (type == NSR_CONTRACT_UN_BOUNDED_SPHERE) {
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] = v[k] * 2.f - 1.f;
        const float mag = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        if (mag > 1.f) {
            const float s = 2.f - 1.f / mag;
#pragma unroll
            for (int k = 0; k < 3; ++k) v[k] = s * (v[k] / mag);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] = v[k] / 4.f + 0.5f;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) x01[3ull * i + k] = v[k];
}

__device__ __forceinline__ bool wave_ray(const int32_t *__restrict__ packed, uint32_t n_rays, uint32_t &r,
                                         uint32_t &start, uint32_t &count)
{
    r = blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
    if (r >= n_rays) return false;
    start = (uint32_t)packed[2ull * r];
    count = (uint32_t)packed[2ull * r + 1];
    return true;
}

// kept[r] = #{ i in ray r : T_i >= eps },  T_i = prod_{j<i} (1 - alpha_j),  alpha = 1 - exp(-exp(logit+bias) * dt)
__global__ void __launch_bounds__(R_BLOCK)
k_visibility_prefix(const __half *__restrict__ mlp_out, uint32_t stride, float bias, const float *__restrict__ t0,
                    const float *__restrict__ t1, const int32_t *__restrict__ packed, float eps,
                    int32_t *__restrict__ kept, uint32_t n_rays)
{
    uint32_t r, start, count;
    if (!wave_ray(packed, n_rays, r, start, count)) return;
    const uint32_t lane = threadIdx.x & 63;
    float carry = 1.f;
    uint32_t n_kept = 0;
    for (uint32_t c = 0; c < count; c += 64) {
        const uint32_t k = c + lane;
        const bool ok = k < count;
        float one_minus_alpha = 1.f;
        if (ok) {
            const float sigma = expf(__half2float(mlp_out[(uint64_t)(start + k) * stride]) + bias);
            const float alpha = 1.f - expf(-sigma * (t1[start + k] - t0[start + k]));
            one_minus_alpha = 1.f - alpha;
        }
        const float inc = wave_incl_scan_mul(one_minus_alpha);
        float exc = __shfl_up(inc, 1, 64);
        if (lane == 0) exc = 1.f;
        const float T = carry * exc;
        const unsigned long long m = __ballot(ok && T >= eps);
        n_kept += (uint32_t)__popcll(m);
        carry *= __shfl(inc, 63, 64);
        if (carry < eps) break;  // wave-uniform: everything after is invisible
    }
    if (lane == 0) kept[r] = (int32_t)n_kept;
}

// The same with one sum per block of VIS_RPB rays beside the counts: the kept-row copy of the main pass forms each ray's offset
// from these (k_copy_kept_rows<.., true>), which takes the one-workgroup scan kernel between the two out of the step's chain.
constexpr int VIS_BLOCK = 512;
constexpr int VIS_RPB = VIS_BLOCK / NSR_WAVE;  // 8 rays per block
__global__ void __launch_bounds__(VIS_BLOCK)
k_visibility_prefix_sums(const __half *__restrict__ mlp_out, uint32_t stride, float bias, const float *__restrict__ t0,
                         const float *__restrict__ t1, const int32_t *__restrict__ packed, float eps,
                         int32_t *__restrict__ kept, int32_t *__restrict__ block_sums, uint32_t n_rays)
{
    __shared__ int32_t sh[VIS_RPB];
    const uint32_t r = blockIdx.x * VIS_RPB + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    uint32_t n_kept = 0;
    if (r < n_rays) {
        const uint32_t start = (uint32_t)packed[2ull * r], count = (uint32_t)packed[2ull * r + 1];
        float carry = 1.f;
        for (uint32_t c = 0; c < count; c += 64) {
            const uint32_t k = c + lane;
            const bool ok = k < count;
            float one_minus_alpha = 1.f;
            if (ok) {
                const float sigma = expf(__half2float(mlp_out[(uint64_t)(start + k) * stride]) + bias);
                const float alpha = 1.f - expf(-sigma * (t1[start + k] - t0[start + k]));
                one_minus_alpha = 1.f - alpha;
            }
            const float inc = wave_incl_scan_mul(one_minus_alpha);
            float exc = __shfl_up(inc, 1, 64);
            if (lane == 0) exc = 1.f;
            const float T = carry * exc;
            const unsigned long long m = __ballot(ok && T >= eps);
            n_kept += (uint32_t)__popcll(m);
            carry *= __shfl(inc, 63, 64);
            if (carry < eps) break;
        }
        if (lane == 0) kept[r] = (int32_t)n_kept;
    }
    if (lane == 0) sh[threadIdx.x >> 6] = (int32_t)n_kept;
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t t = 0;
#pragma unroll
        for (int w = 0; w < VIS_RPB; ++w) t += sh[w];
        block_sums[blockIdx.x] = t;
    }
}

__global__ void __launch_bounds__(R_BLOCK)
k_copy_ray_prefixes(const int32_t *__restrict__ packed_old, const int32_t *__restrict__ packed_new,
                    const float *__restrict__ t0, const float *__restrict__ t1, int64_t *__restrict__ ri_o,
                    float *__restrict__ t0_o, float *__restrict__ t1_o, uint32_t n_rays)
{
    const uint32_t r = blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= n_rays) return;
    const uint32_t src = (uint32_t)packed_old[2ull * r];
    const uint32_t dst = (uint32_t)packed_new[2ull * r], cnt = (uint32_t)packed_new[2ull * r + 1];
    for (uint32_t k = lane; k < cnt; k += 64) {
        t0_o[dst + k] = t0[src + k];
        t1_o[dst + k] = t1[src + k];
        ri_o[dst + k] = (int64_t)r;
    }
}

// SH degree 4 of a unit direction as the texture network sees it: dirs01 = (d + 1)/2 ; SH maps back 2u - 1 (two
// roundings each way, exactly as texture.py:24 + tcnn do) -> 16 halfs
__device__ __forceinline__ void sh4_of_dir(float d0, float d1, float d2, __half2 (&h)[8])
{
    const float u0 = (d0 + 1.f) / 2.f, u1 = (d1 + 1.f) / 2.f, u2 = (d2 + 1.f) / 2.f;
    const float x = u0 * 2.f - 1.f, y = u1 * 2.f - 1.f, z = u2 * 2.f - 1.f;
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    float o[16];
    o[0] = 0.28209479177387814f;
    o[1] = -0.48860251190291987f * y;
    o[2] = 0.48860251190291987f * z;
    o[3] = -0.48860251190291987f * x;
    o[4] = 1.0925484305920792f * xy;
    o[5] = -1.0925484305920792f * yz;
    o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
    o[7] = -1.0925484305920792f * xz;
    o[8] = 0.54627421529603959f * (x2 - y2);
    o[9] = 0.59004358992664352f * y * (-3.f * x2 + y2);
    o[10] = 2.8906114426405538f * xy * z;
    o[11] = 0.45704579946446572f * y * (1.f - 5.f * z2);
    o[12] = 0.3731763325901154f * z * (5.f * z2 - 3.f);
    o[13] = 0.45704579946446572f * x * (1.f - 5.f * z2);
    o[14] = 1.4453057213202769f * z * (x2 - y2);
    o[15] = 0.59004358992664352f * x * (-x2 + 3.f * y2);
#pragma unroll
    for (int k = 0; k < 8; ++k) h[k] = __floats2half2_rn(o[2 * k], o[2 * k + 1]);
}

// Row-wise variant: the kept prefix of a ray is ONE contiguous block in both the marched and the pruned layout, so
// carrying per-sample rows (encoded features, MLP activations, MLP outputs, positions) over the pruning step is a
// per-ray memcpy -- the main pass then reuses what the sigma pass already computed instead of re-encoding.
struct RowCopy { const uint32_t *src; uint32_t *dst; uint32_t row_dwords; uint32_t planes; uint64_t src_plane, dst_plane; };
struct RowCopies { RowCopy a[8]; uint32_t n; };

__global__ void __launch_bounds__(R_BLOCK)
k_copy_ray_prefix_rows(const int32_t *__restrict__ packed_old, const int32_t *__restrict__ packed_new, const RowCopies rc,
                       const float *__restrict__ rays_d, float *__restrict__ dirs_out, int64_t *__restrict__ ri_o,
                       const __half *__restrict__ tex_src, uint32_t tex_src_stride, __half *__restrict__ tex_in,
                       uint32_t n_rays)
{
    const uint32_t r = blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= n_rays) return;
    const uint32_t src = (uint32_t)packed_old[2ull * r];
    const uint32_t dst = (uint32_t)packed_new[2ull * r], cnt = (uint32_t)packed_new[2ull * r + 1];
    if (cnt == 0) return;
    for (uint32_t q = 0; q < rc.n; ++q) {
        const uint32_t rd = rc.a[q].row_dwords, planes = rc.a[q].planes;
        const uint32_t nd = cnt * rd;  // dwords of this ray in one plane
        if (planes == 1) {
            const uint32_t *s = rc.a[q].src + (uint64_t)src * rd;
            uint32_t *d = rc.a[q].dst + (uint64_t)dst * rd;
            if ((rd & 3u) == 0) {  // rows are multiples of 16 B (and the bases 256-B aligned): 1 KiB per wave-instruction
                const uint4 *s4 = reinterpret_cast<const uint4 *>(s);
                uint4 *d4 = reinterpret_cast<uint4 *>(d);
                for (uint32_t w = lane; w < (nd >> 2); w += 64) d4[w] = s4[w];
            } else {
                for (uint32_t w = lane; w < nd; w += 64) d[w] = s[w];
            }
        } else {
            // level-major array = `planes` arrays of rows: the lanes run over (plane, dword) jointly -- a ray keeps ~13
            // samples, one loop per plane would leave 50 of 64 lanes idle 16 times over
            const float inv = 1.f / (float)nd;
            for (uint32_t w = lane; w < planes * nd; w += 64) {
                uint32_t pl = (uint32_t)(((float)w + 0.5f) * inv);  // w / nd (w < 2^22: exact up to the fix-up below)
                uint32_t e = w - pl * nd;
                if (e >= nd) { e += nd; pl -= 1; }  // e wrapped negative: the estimate was one too high
                rc.a[q].dst[pl * rc.a[q].dst_plane + (uint64_t)dst * rd + e] =
                    rc.a[q].src[pl * rc.a[q].src_plane + (uint64_t)src * rd + e];
            }
        }
    }
    const float d0 = rays_d ? rays_d[3ull * r] : 0.f, d1 = rays_d ? rays_d[3ull * r + 1] : 0.f,
                d2 = rays_d ? rays_d[3ull * r + 2] : 0.f;
    if (dirs_out) {
        for (uint32_t k = lane; k < cnt; k += 64) {
            dirs_out[3ull * (dst + k)] = d0; dirs_out[3ull * (dst + k) + 1] = d1; dirs_out[3ull * (dst + k) + 2] = d2;
        }
    }
    if (ri_o)
        for (uint32_t k = lane; k < cnt; k += 64) ri_o[dst + k] = (int64_t)r;
    if (tex_in) {  // texture-network input [16 features | SH4(dir)]: the direction -- hence the SH half -- is per RAY
        __half2 h[8];
        sh4_of_dir(d0, d1, d2, h);
        for (uint32_t k = lane; k < cnt; k += 64) {
            const uint4 *f = reinterpret_cast<const uint4 *>(tex_src + (uint64_t)(src + k) * tex_src_stride);
            uint4 *o = reinterpret_cast<uint4 *>(tex_in + (uint64_t)(dst + k) * 32);
            o[0] = f[0];
            o[1] = f[1];
            o[2] = *reinterpret_cast<uint4 *>(&h[0]);
            o[3] = *reinterpret_cast<uint4 *>(&h[4]);
        }
    }
}

// The main pass's copy, specialised for its row set (t0, t1: 1 dword; x01: 3; level-major encoding with F = 2: `planes`
// planes of 1 dword; density-MLP output: 32 B; NH saved activation rows: 128 B each): ONE LANE PER KEPT SAMPLE issues the
// loads of all its rows before the first store, so a ray costs one memory round trip instead of one per array -- a ray
// keeps ~13 samples, the copy is latency-, not bandwidth-bound.  Also writes ray_indices and the texture input.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // a NATIVE 16-byte vector: arrays of HIP's uint4 (a struct
                                                             // around a union) are not promoted to registers
struct KeptRows {
    const float *t0_s, *t1_s, *x01_s;
    const uint32_t *enc_s;
    const u32x4 *out1_s, *acts_s;
    float *t0_d, *t1_d, *x01_d;
    uint32_t *enc_d;
    u32x4 *out1_d, *acts_d;
    uint64_t enc_sp, enc_dp;  // plane strides (dwords)
    uint64_t acts_sl, acts_dl;  // hidden-layer strides (uint4 units)
    uint32_t planes;
};

// PLANES16: exactly 16 encoding planes (every reference config: 16 levels x 2 features) -- the body is then straight-line
// code and every row stays in registers; with the generic plane loop in between, the compiler kept the activation rows in
// SCRATCH (an un-promoted 128-byte array per hidden layer: 8 + 8 scratch round trips per sample on the step's critical path)
// SCAN: packed_new does not exist yet -- the wave forms its ray's offset itself from the per-ray kept counts and the sums
// k_visibility_prefix_sums left per block of VIS_RPB rays (<= 8,192 of them: sixteen 16-byte loads per lane at most, all L2
// hits), clamps it to the capacity exactly as k_pack_from_counts does, and WRITES packed_new[r]; the wave of the last ray also
// publishes the total and the packing statistics.
struct KeptScan {
    const int32_t *kept, *block_sums;
    int32_t *total, *stats;
    uint32_t capacity;
};

template <int NH, bool PLANES16, bool SCAN>
__global__ void __launch_bounds__(R_BLOCK)
k_copy_kept_rows(const int32_t *__restrict__ packed_old, int32_t *__restrict__ packed_new, const KeptRows kr,
                 const float *__restrict__ rays_d, int64_t *__restrict__ ri_o, __half *__restrict__ tex_in,
                 uint32_t n_rays, const KeptScan ks)
{
    const uint32_t r = blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= n_rays) return;
    const uint32_t src = (uint32_t)packed_old[2ull * r];
    uint32_t dst, cnt;
    if constexpr (SCAN) {
        const uint32_t nb = r / VIS_RPB;  // whole blocks of rays in front of r
        int32_t part = 0;
        for (uint32_t b = lane * 4u; b < nb; b += 256u) {
            if (b + 4u <= nb) {
                const int4 v = *reinterpret_cast<const int4 *>(ks.block_sums + b);
                part += (v.x + v.y) + (v.z + v.w);
            } else {
                for (uint32_t q = b; q < nb; ++q) part += ks.block_sums[q];
            }
        }
        const uint32_t rr = nb * VIS_RPB + lane;  // the rays of r's own block in front of it
        if (rr < r) part += ks.kept[rr];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
        const int32_t own = ks.kept[r];
        int32_t start = part, c = own;
        if (ks.capacity) {  // fixed-size sample buffers: rays past the capacity are truncated (and reported)
            start = min(start, (int32_t)ks.capacity);
            c = min(c, (int32_t)ks.capacity - start);
        }
        if (lane == 0) {
            packed_new[2ull * r] = start;
            packed_new[2ull * r + 1] = c;
            if (r == n_rays - 1u) {
                const int32_t t = part + own;
                ks.total[0] = ks.capacity ? min(t, (int32_t)ks.capacity) : t;
                if (ks.stats) {  // (as k_pack_from_counts)
                    atomicAdd(reinterpret_cast<unsigned long long *>(ks.stats + 4), (unsigned long long)t);
                    atomicMax(&ks.stats[1], t);
                    if (ks.capacity && t > (int32_t)ks.capacity) atomicAdd(&ks.stats[2], 1);
                    ks.stats[0] = t;
                }
            }
        }
        dst = (uint32_t)start;
        cnt = (uint32_t)c;
    } else {
        dst = (uint32_t)packed_new[2ull * r];
        cnt = (uint32_t)packed_new[2ull * r + 1];
    }
    if (cnt == 0) return;
    u32x4 sh_lo, sh_hi;  // (by value: punning an array of half2 through a pointer would put it -- and the rows below -- in scratch)
    {
        __half2 sh[8];
        sh4_of_dir(rays_d[3ull * r], rays_d[3ull * r + 1], rays_d[3ull * r + 2], sh);
        uint32_t w[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) w[q] = __builtin_bit_cast(uint32_t, sh[q]);
        sh_lo = u32x4{w[0], w[1], w[2], w[3]};
        sh_hi = u32x4{w[4], w[5], w[6], w[7]};
    }
    for (uint32_t k = lane; k < cnt; k += 64) {
        const uint64_t i = src + k, o = dst + k;
        const float a0 = kr.t0_s[i], a1 = kr.t1_s[i];
        const float p0 = kr.x01_s[3 * i], p1 = kr.x01_s[3 * i + 1], p2 = kr.x01_s[3 * i + 2];
        const u32x4 f0 = kr.out1_s[2 * i], f1 = kr.out1_s[2 * i + 1];
        u32x4 act[NH][8];
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int j = 0; j < 8; ++j) act[h][j] = kr.acts_s[h * kr.acts_sl + 8 * i + j];
        if constexpr (PLANES16) {
            uint32_t e[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) e[j] = kr.enc_s[j * kr.enc_sp + i];
#pragma unroll
            for (int j = 0; j < 16; ++j) kr.enc_d[j * kr.enc_dp + o] = e[j];
        } else {
            for (uint32_t pb = 0; pb < kr.planes; pb += 16) {  // 16 encoding planes in flight per pass
                uint32_t e[16];
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (pb + j < kr.planes) e[j] = kr.enc_s[(pb + j) * kr.enc_sp + i];
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (pb + j < kr.planes) kr.enc_d[(pb + j) * kr.enc_dp + o] = e[j];
            }
        }
        kr.t0_d[o] = a0;
        kr.t1_d[o] = a1;
        kr.x01_d[3 * o] = p0; kr.x01_d[3 * o + 1] = p1; kr.x01_d[3 * o + 2] = p2;
        kr.out1_d[2 * o] = f0; kr.out1_d[2 * o + 1] = f1;
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int j = 0; j < 8; ++j) kr.acts_d[h * kr.acts_dl + 8 * o + j] = act[h][j];
        ri_o[o] = (int64_t)r;
        u32x4 *t = reinterpret_cast<u32x4 *>(tex_in + o * 32);
        t[0] = f0;
        t[1] = f1;
        t[2] = sh_lo;
        t[3] = sh_hi;
    }
}

// tex_in[n,32] half = [ mlp_out[:, :16] | SH4((d+1)/2) ]  (the fp16 feature IS what .float() then fp16-cast returns)
__global__ void __launch_bounds__(EW_BLOCK)
k_texture_input(const __half *__restrict__ mlp_out, uint32_t stride, const float *__restrict__ dirs,
                __half *__restrict__ tex_in, uint32_t n, const int32_t *__restrict__ n_dev)
{
    const uint32_t i = blockIdx.x * EW_BLOCK + threadIdx.x;
    if (i >= live_count(n, n_dev)) return;
    const uint4 *src = reinterpret_cast<const uint4 *>(mlp_out + (uint64_t)i * stride);
    uint4 *dst = reinterpret_cast<uint4 *>(tex_in + (uint64_t)i * 32);
    dst[0] = src[0];
    dst[1] = src[1];
    __half2 h[8];
    sh4_of_dir(dirs[3ull * i], dirs[3ull * i + 1], dirs[3ull * i + 2], h);
    dst[2] = *reinterpret_cast<uint4 *>(&h[0]);
    dst[3] = *reinterpret_cast<uint4 *>(&h[4]);
}

// per ray: sigma = exp(logit + bias); w_i = T_i (1 - exp(-sigma_i dt_i)); comp_rgb = sum w rgb + bg (1 - sum w)
__global__ void __launch_bounds__(R_BLOCK)
k_composite_forward(const __half *__restrict__ mlp_out, uint32_t stride, float bias, const float *__restrict__ t0,
                    const float *__restrict__ t1, const __half *__restrict__ rgb, uint32_t rgb_stride,
                    const int32_t *__restrict__ packed, const float *__restrict__ bg, float *__restrict__ weights,
                    float *__restrict__ trans, float *__restrict__ comp_rgb, float *__restrict__ opacity,
                    float *__restrict__ depth, uint32_t n_rays,
                    const float *__restrict__ l1_gt /* with l1_part: the masked smooth-L1 loss against these colours ... */,
                    float *__restrict__ l1_part /* ... as one partial (sum, valid rays) per block: [2][gridDim.x] */)
{
    uint32_t r, start, count;
    const bool active = wave_ray(packed, n_rays, r, start, count);
    if (!active) {
        if (!l1_part) return;
        count = 0;  // (an idle wave of the last block still meets the others at the block's partial sum)
    }
    const uint32_t lane = threadIdx.x & 63;
    float carry = 0.f;  // running sum of sigma*dt
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};  // opacity, depth, r, g, b
    for (uint32_t c = 0; c < count; c += 64) {
        const uint32_t k = c + lane;
        const bool ok = k < count;
        float sd = 0.f, a = 0.f, mid = 0.f;
        if (ok) {
            const float ts = t0[start + k], te = t1[start + k];
            const float sigma = expf(__half2float(mlp_out[(uint64_t)(start + k) * stride]) + bias);
            sd = sigma * (te - ts);
            a = 1.f - expf(-sd);
            mid = (ts + te) / 2.f;
        }
        const float inc = wave_incl_scan_add(sd);
        // exclusive prefix by shuffle, not as `inc - sd`: an overflowed density (exp(logit) = inf) would give inf - inf = NaN,
        // where nerfacc's sequential loop gives T = 0 behind the sample (seen as a NaN pixel in an eval render)
        float exc = __shfl_up(inc, 1, 64);
        if (lane == 0) exc = 0.f;
        const float T = expf(-(carry + exc));
        const float w = T * a;
        if (ok) {
            weights[start + k] = w;
            trans[start + k] = T;
            const __half *c3 = rgb + (uint64_t)(start + k) * rgb_stride;
            acc[0] += w;
            acc[1] += w * mid;
            acc[2] += w * __half2float(c3[0]);
            acc[3] += w * __half2float(c3[1]);
            acc[4] += w * __half2float(c3[2]);
        }
        carry += __shfl(inc, 63, 64);
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) acc[q] = wave_sum(acc[q]);
    float l1_s = 0.f, l1_c = 0.f;
    if (active && lane == 0) {
        opacity[r] = acc[0];
        depth[r] = acc[1];
        const float rest = 1.f - acc[0];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float v = acc[2 + q] + bg[q] * rest;
            comp_rgb[3ull * r + q] = v;
            if (l1_part && acc[0] > 0.f) {  // (the arithmetic of k_smooth_l1_valid_set on the value just stored)
                const float d = fabsf(v - l1_gt[3ull * r + q]);
                l1_s += d < 1.f ? 0.5f * d * d : d - 0.5f;
            }
        }
        if (l1_part && acc[0] > 0.f) l1_c = 1.f;
    }
    if (l1_part) {
        // loss sum and valid-ray count of this block's rays: one plain store per block, summed (in a fixed order) by every
        // block of the backward kernel -- no atomics, no zeroing, no one-workgroup reduction kernel between the two
        __shared__ float sh[2][RAYS_PER_BLOCK];
        if (lane == 0) { sh[0][threadIdx.x >> 6] = l1_s; sh[1][threadIdx.x >> 6] = l1_c; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float ts = 0.f, tc = 0.f;
#pragma unroll
            for (int w = 0; w < RAYS_PER_BLOCK; ++w) { ts += sh[0][w]; tc += sh[1][w]; }
            l1_part[blockIdx.x] = ts;
            l1_part[gridDim.x + blockIdx.x] = tc;
        }
    }
}

// loss = mean over valid rays x 3 channels of smooth_l1(comp - gt), valid = opacity > 0
__global__ void __launch_bounds__(EW_BLOCK)
k_smooth_l1_valid(const float *__restrict__ comp_rgb, const float *__restrict__ opacity, const float *__restrict__ gt,
                  float *__restrict__ acc /* [0]=sum, [1]=valid rays */, uint32_t n_rays)
{
    const uint32_t r = blockIdx.x * EW_BLOCK + threadIdx.x;
    float s = 0.f, c = 0.f;
    if (r < n_rays && opacity[r] > 0.f) {
        c = 1.f;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float d = fabsf(comp_rgb[3ull * r + q] - gt[3ull * r + q]);
            s += d < 1.f ? 0.5f * d * d : d - 0.5f;
        }
    }
    s = wave_sum(s);
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0 && c > 0.f) {
        unsafeAtomicAdd(acc, s);
        unsafeAtomicAdd(acc + 1, c);
    }
}

// The same sums by ONE workgroup, written (not accumulated): the 2 x (rays / 64) same-address float atomics of the
// kernel above cost ~17 us at 8192 rays, and the caller no longer has to zero acc first.
__global__ void __launch_bounds__(1024)
k_smooth_l1_valid_set(const float *__restrict__ comp_rgb, const float *__restrict__ opacity, const float *__restrict__ gt,
                      float *__restrict__ acc, uint32_t n_rays)
{
    __shared__ float part[2][16];
    float s = 0.f, c = 0.f;
    for (uint32_t r = threadIdx.x; r < n_rays; r += 1024) {
        if (opacity[r] > 0.f) {
            c += 1.f;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const float d = fabsf(comp_rgb[3ull * r + q] - gt[3ull * r + q]);
                s += d < 1.f ? 0.5f * d * d : d - 0.5f;
            }
        }
    }
    s = wave_sum(s);
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = s; part[1][threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float ts = 0.f, tc = 0.f;
        for (int w = 0; w < 16; ++w) { ts += part[0][w]; tc += part[1][w]; }
        acc[0] = ts;
        acc[1] = tc;
    }
}

__global__ void __launch_bounds__(EW_BLOCK)
k_smooth_l1_valid_bwd(const float *__restrict__ comp_rgb, const float *__restrict__ opacity,
                      const float *__restrict__ gt, const float *__restrict__ acc, float scale,
                      float *__restrict__ g_comp, uint32_t n_rays)
{
    const uint32_t r = blockIdx.x * EW_BLOCK + threadIdx.x;
    if (r >= n_rays) return;
    const bool valid = opacity[r] > 0.f;
    const float inv = scale / fmaxf(3.f * acc[1], 1.f);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const float d = comp_rgb[3ull * r + q] - gt[3ull * r + q];
        const float g = fabsf(d) < 1.f ? d : (d > 0.f ? 1.f : -1.f);
        g_comp[3ull * r + q] = valid ? g * inv : 0.f;
    }
}

// backward of composite_forward w.r.t. rgb and the density logit (trunc_exp backward folded in)
__global__ void __launch_bounds__(R_BLOCK)
k_composite_backward(const __half *__restrict__ mlp_out, uint32_t stride, float bias, const float *__restrict__ t0,
                     const float *__restrict__ t1, const __half *__restrict__ rgb, uint32_t rgb_stride,
                     const int32_t *__restrict__ packed, const float *__restrict__ bg,
                     const float *__restrict__ weights, const float *__restrict__ trans,
                     const float *__restrict__ g_comp, const float *__restrict__ g_opacity,
                     const float *__restrict__ g_depth, float *__restrict__ d_rgb, float *__restrict__ d_logit,
                     uint32_t n_rays, const float *__restrict__ l1_comp, const float *__restrict__ l1_opacity,
                     const float *__restrict__ l1_gt, const float *__restrict__ l1_acc, float l1_scale,
                     const float *__restrict__ g_weights /* dL/d weights[n] of the caller's own loss terms, or NULL */,
                     const float *__restrict__ l1_part /* [2][gridDim.x] block partials of k_composite_forward, or NULL */,
                     float *__restrict__ l1_acc_out /* with l1_part: (loss sum, valid rays) for the caller, written by block 0 */)
{
    const uint32_t lane = threadIdx.x & 63;
    float n_valid = 0.f;
    if (l1_part) {  // every block sums the forward's partials itself (gridDim.x floats x 2, L2-resident), same order everywhere
        __shared__ float tot[2][R_BLOCK / 64];
        float s = 0.f, c = 0.f;
        for (uint32_t k = threadIdx.x; k < gridDim.x; k += R_BLOCK) { s += l1_part[k]; c += l1_part[gridDim.x + k]; }
        s = wave_sum(s);
        c = wave_sum(c);
        if (lane == 0) { tot[0][threadIdx.x >> 6] = s; tot[1][threadIdx.x >> 6] = c; }
        __syncthreads();
        s = c = 0.f;
#pragma unroll
        for (int w = 0; w < R_BLOCK / 64; ++w) { s += tot[0][w]; c += tot[1][w]; }
        n_valid = c;
        if (blockIdx.x == 0 && threadIdx.x == 0) { l1_acc_out[0] = s; l1_acc_out[1] = c; }
    }
    uint32_t r, start, count;
    if (!wave_ray(packed, n_rays, r, start, count)) return;
    float g0, g1, g2;
    if (l1_comp) {  // gradient of the masked smooth-L1 loss evaluated here (k_smooth_l1_valid_bwd without its launch)
        const bool valid = l1_opacity[r] > 0.f;
        const float inv = l1_scale / fmaxf(3.f * (l1_part ? n_valid : l1_acc[1]), 1.f);
        float g[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float d = l1_comp[3ull * r + q] - l1_gt[3ull * r + q];
            const float gq = fabsf(d) < 1.f ? d : (d > 0.f ? 1.f : -1.f);
            g[q] = valid ? gq * inv : 0.f;
        }
        g0 = g[0]; g1 = g[1]; g2 = g[2];
    } else {
        g0 = g_comp[3ull * r]; g1 = g_comp[3ull * r + 1]; g2 = g_comp[3ull * r + 2];
    }
    const float b0 = bg[0], b1 = bg[1], b2 = bg[2];
    const float gop = g_opacity ? g_opacity[r] : 0.f, gdp = g_depth ? g_depth[r] : 0.f;
    float carry = 0.f;  // sum_{i > j} gT_i T_i, walking the ray from its end
    for (uint32_t c = 0; c < count; c += 64) {
        const uint32_t k = c + lane;
        const bool ok = k < count;
        const uint32_t idx = start + count - 1 - k;
        float v = 0.f, gw = 0.f, T = 0.f, a = 0.f, dt = 0.f, z = 0.f;
        if (ok) {
            const float ts = t0[idx], te = t1[idx];
            dt = te - ts;
            z = __half2float(mlp_out[(uint64_t)idx * stride]) + bias;
            const float sd = expf(z) * dt;
            a = 1.f - expf(-sd);
            T = trans[idx];
            const __half *c3 = rgb + (uint64_t)idx * rgb_stride;
            const float w = weights[idx];
            gw = g0 * (__half2float(c3[0]) - b0) + g1 * (__half2float(c3[1]) - b1) + g2 * (__half2float(c3[2]) - b2) +
                 gop + gdp * ((ts + te) / 2.f) + (g_weights ? g_weights[idx] : 0.f);
            d_rgb[3ull * idx] = w * g0;
            d_rgb[3ull * idx + 1] = w * g1;
            d_rgb[3ull * idx + 2] = w * g2;
            v = gw * a * T;  // gT_i * T_i
        }
        const float inc = wave_incl_scan_add(v);
        if (ok) {
            const float g_sd = gw * T * (1.f - a) - (carry + (inc - v));
            d_logit[idx] = g_sd * dt * expf(fminf(z, 15.f));
        }
        carry += __shfl(inc, 63, 64);
    }
}

// segmented wave scans of the compositing kernels below: DPP row shifts + row broadcasts, registers only
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float flat_dpp(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
// inclusive segmented sum over the wave; dist = lane - (first lane of this lane's segment inside the wave)
__device__ __forceinline__ float flat_seg_scan(float v, uint32_t dist, uint32_t lane)
{
    const uint32_t in_row = lane & 15u;
    float t;
    t = flat_dpp<0x111, 0xf>(v); if (dist >= 1u && in_row >= 1u) v += t;
    t = flat_dpp<0x112, 0xf>(v); if (dist >= 2u && in_row >= 2u) v += t;
    t = flat_dpp<0x114, 0xf>(v); if (dist >= 4u && in_row >= 4u) v += t;
    t = flat_dpp<0x118, 0xf>(v); if (dist >= 8u && in_row >= 8u) v += t;
    t = flat_dpp<0x142, 0xa>(v); if ((lane & 16u) && dist > in_row) v += t;           // row_bcast:15 into rows 1, 3
    t = flat_dpp<0x143, 0xc>(v); if (lane >= 32u && dist >= lane - 31u) v += t;       // row_bcast:31 into rows 2, 3
    return v;
}

// ---- sample-partitioned compositing (round 6) --------------------------------------------------------------------------------
// The wave-per-ray kernels above leave > 80 % of their lanes idle at the step's operating point (8,192 ray slots, ~1e5 kept
// samples: a ray keeps 10-15 samples, two thirds of the slots keep none); round 5's flat kernels (a wave per FOUR rays, walking
// their samples 64 at a time, serially) had a tail -- most waves one under-filled chunk, a few 5-10 dependent ones -- and are
// gone (profiles/r06_step_variants_compositing.json: same-process A/B in the step).  Here a wave owns 64
// consecutive SAMPLES of the packed arrays whatever the ray boundaries (ray id from the kept rows' ray_indices, its segment
// from packed_info).  What a chunk needs from outside is the state of the ONE ray that is open at its first lane: the wave
// recomputes it itself from that ray's earlier samples (forward: sum of sigma dt and the five weighted sums; backward: the
// suffix sum of gT T behind the chunk) -- ~18 samples on average, one extra 64-lane pass that overlaps the chunk's own loads.
// No cross-wave dependency, no atomics, every wave does a bounded amount of work, every sum in a fixed order.
constexpr int SP_BLOCK = 256;

struct SpSample { float sd, a, mid, cr, cg, cb; };
__device__ __forceinline__ SpSample sp_load(const __half *__restrict__ mlp_out, uint32_t stride, float bias,
                                            const float *__restrict__ t0, const float *__restrict__ t1,
                                            const __half *__restrict__ rgb, uint32_t rgb_stride, uint32_t i)
{
    SpSample s;
    const float ts = t0[i], te = t1[i];
    const float sigma = expf(__half2float(mlp_out[(uint64_t)i * stride]) + bias);
    s.sd = sigma * (te - ts);
    s.a = 1.f - expf(-s.sd);
    s.mid = (ts + te) / 2.f;
    const __half *c3 = rgb + (uint64_t)i * rgb_stride;
    s.cr = __half2float(c3[0]); s.cg = __half2float(c3[1]); s.cb = __half2float(c3[2]);
    return s;
}

__global__ void __launch_bounds__(SP_BLOCK)
k_composite_forward_samples(const __half *__restrict__ mlp_out, uint32_t stride, float bias, const float *__restrict__ t0,
                            const float *__restrict__ t1, const __half *__restrict__ rgb, uint32_t rgb_stride,
                            const int32_t *__restrict__ packed, const int64_t *__restrict__ ray_idx,
                            const float *__restrict__ bg, float *__restrict__ weights, float *__restrict__ trans,
                            float *__restrict__ comp_rgb, float *__restrict__ opacity, float *__restrict__ depth,
                            uint32_t n_rays, uint32_t n_samples, const int32_t *__restrict__ n_dev,
                            const float *__restrict__ l1_gt, float *__restrict__ l1_part /* [2][gridDim.x] or NULL */)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t n_live = live_count(n_samples, n_dev);
    const float b0 = bg[0], b1 = bg[1], b2 = bg[2];
    // rays without samples: background, outside the loss (opacity 0)
    for (uint32_t r = blockIdx.x * SP_BLOCK + threadIdx.x; r < n_rays; r += gridDim.x * SP_BLOCK)
        if (packed[2ull * r + 1] == 0) {
            opacity[r] = 0.f;
            depth[r] = 0.f;
            comp_rgb[3ull * r] = b0; comp_rgb[3ull * r + 1] = b1; comp_rgb[3ull * r + 2] = b2;
        }
    float l1_s = 0.f, l1_c = 0.f;
    for (uint32_t c0 = (blockIdx.x * (SP_BLOCK / 64) + wave) * 64u; c0 < n_live; c0 += gridDim.x * SP_BLOCK) {
        const uint32_t i = c0 + lane;
        const bool ok = i < n_live;
        const uint32_t ii = ok ? i : n_live - 1u;
        const uint32_t r = (uint32_t)ray_idx[ii];
        const uint32_t rs = (uint32_t)packed[2ull * r], rc = (uint32_t)packed[2ull * r + 1];
        const uint32_t k = ii - rs;                            // position inside the ray
        const bool open = k > lane;                            // the ray began in front of this chunk
        const uint32_t dist = open ? lane : k;
        SpSample m = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (ok) m = sp_load(mlp_out, stride, bias, t0, t1, rgb, rgb_stride, i);
        // the ray that is open at lane 0: its state in front of the chunk, from its own samples [start, c0)
        float c_sd = 0.f, c_acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        const uint32_t k0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
        for (uint32_t p = c0 - k0; p < c0; p += 64) {
            const uint32_t j = p + lane;
            const bool okj = j < c0;
            SpSample q = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (okj) q = sp_load(mlp_out, stride, bias, t0, t1, rgb, rgb_stride, j);
            const float inc = flat_seg_scan(q.sd, lane, lane);  // (one segment: a plain inclusive scan)
            float exc = __shfl_up(inc, 1, 64);
            if (lane == 0u) exc = 0.f;
            const float T = expf(-(c_sd + exc));
            const float w = okj ? T * q.a : 0.f;
            const float v[5] = {w, w * q.mid, w * q.cr, w * q.cg, w * q.cb};
#pragma unroll
            for (int u = 0; u < 5; ++u) c_acc[u] += __shfl(flat_seg_scan(v[u], lane, lane), 63, 64);
            c_sd += __shfl(inc, 63, 64);
        }
        const float inc = flat_seg_scan(m.sd, dist, lane);
        // exclusive prefix by shift, not as inc - sd: an overflowed density gives inf - inf = NaN (see k_composite_forward)
        float exc = __shfl_up(inc, 1, 64);
        if (dist == 0u) exc = 0.f;
        const float T = expf(-((open ? c_sd : 0.f) + exc));
        const float w = ok ? T * m.a : 0.f;
        float v[5] = {w, w * m.mid, w * m.cr, w * m.cg, w * m.cb};
#pragma unroll
        for (int u = 0; u < 5; ++u) v[u] = flat_seg_scan(v[u], dist, lane) + (open ? c_acc[u] : 0.f);
        if (ok) {
            weights[i] = w;
            trans[i] = T;
            if (k + 1u == rc) {  // last sample of its ray: the sums are complete
                opacity[r] = v[0];
                depth[r] = v[1];
                const float rest = 1.f - v[0];
                const float o3[3] = {v[2] + b0 * rest, v[3] + b1 * rest, v[4] + b2 * rest};
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    comp_rgb[3ull * r + u] = o3[u];
                    if (l1_part && v[0] > 0.f) {
                        const float d = fabsf(o3[u] - l1_gt[3ull * r + u]);
                        l1_s += d < 1.f ? 0.5f * d * d : d - 0.5f;
                    }
                }
                if (l1_part && v[0] > 0.f) l1_c += 1.f;
            }
        }
    }
    if (l1_part) {
        __shared__ float sh[2][SP_BLOCK / 64];
        l1_s = wave_sum(l1_s);
        l1_c = wave_sum(l1_c);
        if (lane == 0) { sh[0][wave] = l1_s; sh[1][wave] = l1_c; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float ts = 0.f, tc = 0.f;
#pragma unroll
            for (int w = 0; w < SP_BLOCK / 64; ++w) { ts += sh[0][w]; tc += sh[1][w]; }
            l1_part[blockIdx.x] = ts;
            l1_part[gridDim.x + blockIdx.x] = tc;
        }
    }
}

// backward of the above w.r.t. rgb and the density logit.  Lane l of a chunk holds sample c0 + 63 - l (suffix sums run from the
// ray's end); the open ray is the one of the chunk's LAST sample, its state = sum of gT T over its samples behind the chunk.
__global__ void __launch_bounds__(SP_BLOCK)
k_composite_backward_samples(const __half *__restrict__ mlp_out, uint32_t stride, float bias, const float *__restrict__ t0,
                             const float *__restrict__ t1, const __half *__restrict__ rgb, uint32_t rgb_stride,
                             const int32_t *__restrict__ packed, const int64_t *__restrict__ ray_idx,
                             const float *__restrict__ bg, const float *__restrict__ weights,
                             const float *__restrict__ trans, const float *__restrict__ g_comp,
                             const float *__restrict__ g_opacity, const float *__restrict__ g_depth,
                             float *__restrict__ d_rgb, float *__restrict__ d_logit, uint32_t n_rays, uint32_t n_samples,
                             const int32_t *__restrict__ n_dev, const float *__restrict__ l1_comp,
                             const float *__restrict__ l1_opacity, const float *__restrict__ l1_gt,
                             const float *__restrict__ l1_acc, float l1_scale, const float *__restrict__ g_weights,
                             const float *__restrict__ l1_part, uint32_t n_part, float *__restrict__ l1_acc_out)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    float n_valid = 0.f;
    if (l1_part) {  // every block sums the forward's partials itself, same order everywhere
        __shared__ float tot[2][SP_BLOCK / 64];
        float s = 0.f, c = 0.f;
        for (uint32_t k = threadIdx.x; k < n_part; k += SP_BLOCK) { s += l1_part[k]; c += l1_part[n_part + k]; }
        s = wave_sum(s);
        c = wave_sum(c);
        if (lane == 0) { tot[0][wave] = s; tot[1][wave] = c; }
        __syncthreads();
        s = c = 0.f;
#pragma unroll
        for (int w = 0; w < SP_BLOCK / 64; ++w) { s += tot[0][w]; c += tot[1][w]; }
        n_valid = c;
        if (blockIdx.x == 0 && threadIdx.x == 0) { l1_acc_out[0] = s; l1_acc_out[1] = c; }
    }
    const uint32_t n_live = live_count(n_samples, n_dev);
    const float b0 = bg[0], b1 = bg[1], b2 = bg[2];
    const float inv = l1_comp ? l1_scale / fmaxf(3.f * (l1_part ? n_valid : l1_acc[1]), 1.f) : 0.f;
    for (uint32_t c0 = (blockIdx.x * (SP_BLOCK / 64) + wave) * 64u; c0 < n_live; c0 += gridDim.x * SP_BLOCK) {
        const uint32_t c1 = min(c0 + 64u, n_live);            // the chunk is [c0, c1)
        const bool ok = lane < c1 - c0;
        const uint32_t i = c1 - 1u - (ok ? lane : 0u);
        const uint32_t r = (uint32_t)ray_idx[i];
        const uint32_t rs = (uint32_t)packed[2ull * r], rc = (uint32_t)packed[2ull * r + 1];
        const uint32_t k = rs + rc - 1u - i;                   // position counted from the ray's end
        const bool open = k > lane;                            // the ray goes on behind this chunk
        const uint32_t dist = open ? lane : k;
        // upstream gradients of this lane's ray
        float g0, g1, g2;
        if (l1_comp) {
            const bool valid = l1_opacity[r] > 0.f;
            float g[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const float d = l1_comp[3ull * r + u] - l1_gt[3ull * r + u];
                const float gq = fabsf(d) < 1.f ? d : (d > 0.f ? 1.f : -1.f);
                g[u] = valid ? gq * inv : 0.f;
            }
            g0 = g[0]; g1 = g[1]; g2 = g[2];
        } else {
            g0 = g_comp[3ull * r]; g1 = g_comp[3ull * r + 1]; g2 = g_comp[3ull * r + 2];
        }
        const float gop = g_opacity ? g_opacity[r] : 0.f, gdp = g_depth ? g_depth[r] : 0.f;
        float v = 0.f, gw = 0.f, T = 0.f, a = 0.f, dt = 0.f, z = 0.f;
        if (ok) {
            const float ts = t0[i], te = t1[i];
            dt = te - ts;
            z = __half2float(mlp_out[(uint64_t)i * stride]) + bias;
            const float sd = expf(z) * dt;
            a = 1.f - expf(-sd);
            T = trans[i];
            const __half *c3 = rgb + (uint64_t)i * rgb_stride;
            const float w = weights[i];
            gw = g0 * (__half2float(c3[0]) - b0) + g1 * (__half2float(c3[1]) - b1) + g2 * (__half2float(c3[2]) - b2) +
                 gop + gdp * ((ts + te) / 2.f) + (g_weights ? g_weights[i] : 0.f);
            d_rgb[3ull * i] = w * g0;
            d_rgb[3ull * i + 1] = w * g1;
            d_rgb[3ull * i + 2] = w * g2;
            v = gw * a * T;  // gT_i * T_i
        }
        // the open ray's samples behind the chunk: [c1, c1 + k0), all of lane 0's ray (same upstream gradients)
        float c_v = 0.f;
        const uint32_t k0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
        const float h0 = __shfl(g0, 0, 64), h1 = __shfl(g1, 0, 64), h2 = __shfl(g2, 0, 64);
        const float hop = __shfl(gop, 0, 64), hdp = __shfl(gdp, 0, 64);
        for (uint32_t p = 0; p < k0; p += 64) {
            const uint32_t j = c1 + p + lane;
            float vj = 0.f;
            if (p + lane < k0) {
                const float ts = t0[j], te = t1[j];
                const float sd = expf(__half2float(mlp_out[(uint64_t)j * stride]) + bias) * (te - ts);
                const float aj = 1.f - expf(-sd);
                const __half *c3 = rgb + (uint64_t)j * rgb_stride;
                const float gwj = h0 * (__half2float(c3[0]) - b0) + h1 * (__half2float(c3[1]) - b1) +
                                  h2 * (__half2float(c3[2]) - b2) + hop + hdp * ((ts + te) / 2.f) +
                                  (g_weights ? g_weights[j] : 0.f);
                vj = gwj * aj * trans[j];
            }
            // summed nearest-first, like the in-chunk suffix scan walks the ray
            c_v += __shfl(flat_seg_scan(vj, lane, lane), 63, 64);
        }
        const float inc = flat_seg_scan(v, dist, lane);
        float exc = __shfl_up(inc, 1, 64);
        if (dist == 0u) exc = 0.f;
        if (ok) {
            const float g_sd = gw * T * (1.f - a) - ((open ? c_v : 0.f) + exc);
            d_logit[i] = g_sd * dt * expf(fminf(z, 15.f));
        }
    }
}

// training-ray gather: pixel (index, y, x) -> ray (o, normalised d), ground-truth rgb blended on the background
__global__ void __launch_bounds__(EW_BLOCK)
k_gather_train_rays(const float *__restrict__ images, const float *__restrict__ masks,
                    const float *__restrict__ directions, const float *__restrict__ c2w,
                    const int64_t *__restrict__ index, const int64_t *__restrict__ px, const int64_t *__restrict__ py,
                    const float *__restrict__ bg, int H, int W, int apply_mask, float *__restrict__ rays,
                    float *__restrict__ rgb, float *__restrict__ fg, uint32_t n)
{
    const uint32_t i = blockIdx.x * EW_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int64_t im = index[i], x = px[i], y = py[i];
    const float *dir = directions + ((size_t)y * W + x) * 3;
    const float *m = c2w + (size_t)im * 12;  // [3,4] row-major
    float d[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) d[k] = (dir[0] * m[4 * k] + dir[1] * m[4 * k + 1]) + dir[2] * m[4 * k + 2];
    const float nrm = fmaxf(sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), 1e-12f);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        rays[6ull * i + k] = m[4 * k + 3];
        rays[6ull * i + 3 + k] = d[k] / nrm;
    }
    const size_t pix = ((size_t)im * H + y) * W + x;
    const float f = masks[pix];
    fg[i] = f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float c = images[pix * 3 + k];
        rgb[3ull * i + k] = apply_mask ? c * f + bg[k] * (1.f - f) : c;
    }
}

// One kernel for everything a training ray needs before marching (systems/nerf.py:38-79 + the slab test and the
// stratified jitter at the top of nerfacc.ray_marching): u01 holds 4 uniform rows [image, x, y, jitter].
__global__ void __launch_bounds__(EW_BLOCK)
k_prepare_train_rays(const float *__restrict__ images, const float *__restrict__ masks,
                     const float *__restrict__ directions, const float *__restrict__ c2w, const float *__restrict__ u01,
                     const float *__restrict__ bg, int n_img, int H, int W, int apply_mask,
                     const float *__restrict__ aabb, float jitter_step, float *__restrict__ rays,
                     float *__restrict__ rays_o, float *__restrict__ rays_d, float *__restrict__ rgb,
                     float *__restrict__ fg, float *__restrict__ t_min, float *__restrict__ t_max, uint32_t n,
                     const int32_t *__restrict__ n_active)
{
    const uint32_t i = blockIdx.x * EW_BLOCK + threadIdx.x;
    if (i >= n) return;
    if (n_active && i >= (uint32_t)*n_active) {
        // slot beyond the current dynamic batch: a DEAD ray (misses the box -> no samples, opacity 0, outside the loss).
        // Keeping the arrays at their maximum size makes the batch size a device-side value: the next marching pass
        // can be queued before the host has seen this step's sample count.
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float dk = k == 2 ? 1.f : 0.f;
            rays[6ull * i + k] = 0.f; rays[6ull * i + 3 + k] = dk;
            rays_o[3ull * i + k] = 0.f; rays_d[3ull * i + k] = dk;
            rgb[3ull * i + k] = 0.f;
        }
        fg[i] = 0.f;
        t_min[i] = 1e10f;
        t_max[i] = 1e10f;
        return;
    }
    const int im = min((int)(u01[i] * (float)n_img), n_img - 1);
    const int x = min((int)(u01[n + i] * (float)W), W - 1);
    const int y = min((int)(u01[2ull * n + i] * (float)H), H - 1);
    const float *dir = directions + ((size_t)y * W + x) * 3;
    const float *m = c2w + (size_t)im * 12;
    float d[3], o[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) d[k] = (dir[0] * m[4 * k] + dir[1] * m[4 * k + 1]) + dir[2] * m[4 * k + 2];
    const float nrm = fmaxf(sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), 1e-12f);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o[k] = m[4 * k + 3];
        d[k] = d[k] / nrm;
        rays[6ull * i + k] = o[k];
        rays[6ull * i + 3 + k] = d[k];
        rays_o[3ull * i + k] = o[k];
        rays_d[3ull * i + k] = d[k];
    }
    const size_t pix = ((size_t)im * H + y) * W + x;
    const float f = masks[pix];
    fg[i] = f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float c = images[pix * 3 + k];
        rgb[3ull * i + k] = apply_mask ? c * f + bg[k] * (1.f - f) : c;
    }
    // slab test: same operation order as csrc/march.hip:aabb_one (bit-exact against the oracle)
    float tmin = (aabb[0] - o[0]) / d[0], tmax = (aabb[3] - o[0]) / d[0];
    if (tmin > tmax) { const float s = tmin; tmin = tmax; tmax = s; }
    float tymin = (aabb[1] - o[1]) / d[1], tymax = (aabb[4] - o[1]) / d[1];
    if (tymin > tymax) { const float s = tymin; tymin = tymax; tymax = s; }
    float near = 1e10f, far = 1e10f;
    if (!(tmin > tymax || tymin > tmax)) {
        if (tymin > tmin) tmin = tymin;
        if (tymax < tmax) tmax = tymax;
        float tzmin = (aabb[2] - o[2]) / d[2], tzmax = (aabb[5] - o[2]) / d[2];
        if (tzmin > tzmax) { const float s = tzmin; tzmin = tzmax; tzmax = s; }
        if (!(tmin > tzmax || tzmin > tmax)) {
            if (tzmin > tmin) tmin = tzmin;
            if (tzmax < tmax) tmax = tzmax;
            near = tmin > 0.f ? tmin : 0.f;
            far = tmax;
        }
    }
    if (jitter_step > 0.f) near = __fadd_rn(near, __fmul_rn(u01[3ull * n + i], jitter_step));  // t_min + rand * step
    t_min[i] = near;
    t_max[i] = far;
}

// systems/nerf.py:93-95 on the device, in the double arithmetic Python uses:
//   t = int(n * (target / S));  n = min(int(n * 0.9 + t * 0.1), max)
__global__ void k_update_ray_count(const int32_t *__restrict__ n_samples, int32_t *__restrict__ n_rays,
                                   int32_t target_samples, int32_t max_rays, long long *__restrict__ rays_accum)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    if (rays_accum) *rays_accum += (long long)*n_rays;  // rays of THIS batch, before the update
    const int32_t s = *n_samples;
    if (s <= 0 || target_samples <= 0) return;
    const double n = (double)*n_rays;
    const double t = (double)(long long)(n * ((double)target_samples / (double)s));
    const long long v = (long long)(n * 0.9 + t * 0.1);
    *n_rays = (int32_t)(v < (long long)max_rays ? v : (long long)max_rays);
}

}  // namespace

#define RAY_GRID(n_rays) dim3(nsr_div_up(n_rays, RAYS_PER_BLOCK)), dim3(R_BLOCK), 0, (hipStream_t)stream
#define EW_GRID(n) dim3(nsr_div_up(n, EW_BLOCK)), dim3(EW_BLOCK), 0, (hipStream_t)stream

extern "C" int nsr_sample_positions_unit(const float *rays_o, const float *rays_d, const int64_t *ray_indices,
                                         const float *t_starts, const float *t_ends, float radius, int contraction,
                                         float *x01, float *dirs_out, uint32_t n, const int32_t *n_dev, void *stream)
{
    NSR_REQUIRE(contraction == NSR_CONTRACT_AABB || contraction == NSR_CONTRACT_UN_BOUNDED_SPHERE,
                "nsr_sample_positions_unit: contraction type %d not implemented", contraction);
    if (n == 0) return NSR_OK;
    NSR_REQUIRE(rays_o && rays_d && ray_indices && t_starts && t_ends && x01, "nsr_sample_positions_unit: NULL pointer");
    hipLaunchKernelGGL(k_sample_positions_unit, EW_GRID(n), rays_o, rays_d, ray_indices, t_starts, t_ends, radius,
                       contraction, x01, dirs_out, n, n_dev);
    NSR_CHECK_LAUNCH("nsr_sample_positions_unit");
    return NSR_OK;
}

extern "C" int nsr_visibility_prefix(const nsr_half *mlp_out, uint32_t stride, float density_bias, const float *t_starts,
                                     const float *t_ends, const int32_t *packed_info, float early_stop_eps,
                                     int32_t *kept_counts, uint32_t n_rays, void *stream)
{
    if (n_rays == 0) return NSR_OK;
    NSR_REQUIRE(packed_info && kept_counts, "nsr_visibility_prefix: NULL pointer");
    hipLaunchKernelGGL(k_visibility_prefix, RAY_GRID(n_rays), (const __half *)mlp_out, stride, density_bias, t_starts,
                       t_ends, packed_info, early_stop_eps, kept_counts, n_rays);
    NSR_CHECK_LAUNCH("nsr_visibility_prefix");
    return NSR_OK;
}

extern "C" int nsr_texture_input(const nsr_half *mlp_out, uint32_t stride, const float *dirs, nsr_half *tex_in,
                                 uint32_t n, const int32_t *n_dev, void *stream)
{
    NSR_REQUIRE(stride >= 16 && (stride & 7u) == 0, "nsr_texture_input: feature rows must be >= 16 halfs, 16-B aligned");
    if (n == 0) return NSR_OK;
    NSR_REQUIRE(mlp_out && dirs && tex_in, "nsr_texture_input: NULL pointer");
    hipLaunchKernelGGL(k_texture_input, EW_GRID(n), (const __half *)mlp_out, stride, dirs, (__half *)tex_in, n, n_dev);
    NSR_CHECK_LAUNCH("nsr_texture_input");
    return NSR_OK;
}

extern "C" int nsr_composite_forward(const nsr_half *mlp_out, uint32_t stride, float density_bias, const float *t_starts,
                                     const float *t_ends, const nsr_half *rgb, uint32_t rgb_stride,
                                     const int32_t *packed_info, const float *background, float *weights, float *trans,
                                     float *comp_rgb, float *opacity, float *depth, uint32_t n_rays, void *stream)
{
    if (n_rays == 0) return NSR_OK;
    NSR_REQUIRE(packed_info && background && comp_rgb && opacity && depth, "nsr_composite_forward: NULL pointer");
    hipLaunchKernelGGL(k_composite_forward, RAY_GRID(n_rays), (const __half *)mlp_out, stride, density_bias, t_starts,
                       t_ends, (const __half *)rgb, rgb_stride, packed_info, background, weights, trans, comp_rgb,
                       opacity, depth, n_rays, nullptr, nullptr);
    NSR_CHECK_LAUNCH("nsr_composite_forward");
    return NSR_OK;
}

extern "C" int nsr_smooth_l1_valid(const float *comp_rgb, const float *opacity, const float *gt_rgb, float *acc2,
                                   uint32_t n_rays, void *stream)
{
    if (n_rays == 0) return NSR_OK;
    NSR_REQUIRE(comp_rgb && opacity && gt_rgb && acc2, "nsr_smooth_l1_valid: NULL pointer");
    hipLaunchKernelGGL(k_smooth_l1_valid, EW_GRID(n_rays), comp_rgb, opacity, gt_rgb, acc2, n_rays);
    NSR_CHECK_LAUNCH("nsr_smooth_l1_valid");
    return NSR_OK;
}

extern "C" int nsr_smooth_l1_valid_set(const float *comp_rgb, const float *opacity, const float *gt_rgb, float *acc2,
                                       uint32_t n_rays, void *stream)
{
    NSR_REQUIRE(acc2 && (n_rays == 0 || (comp_rgb && opacity && gt_rgb)), "nsr_smooth_l1_valid_set: NULL pointer");
    hipLaunchKernelGGL(k_smooth_l1_valid_set, dim3(1), dim3(1024), 0, (hipStream_t)stream, comp_rgb, opacity, gt_rgb,
                       acc2, n_rays);
    NSR_CHECK_LAUNCH("nsr_smooth_l1_valid_set");
    return NSR_OK;
}

extern "C" int nsr_smooth_l1_valid_backward(const float *comp_rgb, const float *opacity, const float *gt_rgb,
                                            const float *acc2, float grad_scale, float *grad_comp_rgb, uint32_t n_rays,
                                            void *stream)
{
    if (n_rays == 0) return NSR_OK;
    NSR_REQUIRE(comp_rgb && opacity && gt_rgb && acc2 && grad_comp_rgb, "nsr_smooth_l1_valid_backward: NULL pointer");
    hipLaunchKernelGGL(k_smooth_l1_valid_bwd, EW_GRID(n_rays), comp_rgb, opacity, gt_rgb, acc2, grad_scale,
                       grad_comp_rgb, n_rays);
    NSR_CHECK_LAUNCH("nsr_smooth_l1_valid_backward");
    return NSR_OK;
}

extern "C" int nsr_composite_backward(const nsr_half *mlp_out, uint32_t stride, float density_bias,
                                      const float *t_starts, const float *t_ends, const nsr_half *rgb,
                                      uint32_t rgb_stride, const int32_t *packed_info, const float *background,
                                      const float *weights, const float *trans, const float *grad_comp_rgb,
                                      const float *grad_opacity, const float *grad_depth, float *grad_rgb,
                                      float *grad_logit, uint32_t n_rays, void *stream)
{
    if (n_rays == 0) return NSR_OK;
    NSR_REQUIRE(packed_info && background && weights && trans && grad_comp_rgb && grad_rgb && grad_logit,
                "nsr_composite_backward: NULL pointer");
    hipLaunchKernelGGL(k_composite_backward, RAY_GRID(n_rays), (const __half *)mlp_out, stride, density_bias, t_starts,
                       t_ends, (const __half *)rgb, rgb_stride, packed_info, background, weights, trans, grad_comp_rgb,
                       grad_opacity, grad_depth, grad_rgb, grad_logit, n_rays, nullptr, nullptr, nullptr, nullptr, 0.f,
                       nullptr, nullptr, nullptr);
    NSR_CHECK_LAUNCH("nsr_composite_backward");
    return NSR_OK;
}

NSR_INTERNAL int nsr_composite_backward_ex(const nsr_half *mlp_out, uint32_t stride, float density_bias, const float *t_starts,
                                         const float *t_ends, const nsr_half *rgb, uint32_t rgb_stride,
                                         const int32_t *packed_info, const float *background, const float *weights,
                                         const float *trans, const float *grad_comp_rgb, const float *grad_opacity,
                                         const float *grad_depth, const float *grad_weights, float *grad_rgb,
                                         float *grad_logit, uint32_t n_rays, void *stream)
{
    if (n_rays == 0) return NSR_OK;
    NSR_REQUIRE(packed_info && background && weights && trans && grad_comp_rgb && grad_rgb && grad_logit,
                "nsr_composite_backward_ex: NULL pointer");
    hipLaunchKernelGGL(k_composite_backward, RAY_GRID(n_rays), (const __half *)mlp_out, stride, density_bias, t_starts,
                       t_ends, (const __half *)rgb, rgb_stride, packed_info, background, weights, trans, grad_comp_rgb,
                       grad_opacity, grad_depth, grad_rgb, grad_logit, n_rays, nullptr, nullptr, nullptr, nullptr, 0.f,
                       grad_weights, nullptr, nullptr);
    NSR_CHECK_LAUNCH("nsr_composite_backward_ex");
    return NSR_OK;
}

extern "C" int nsr_composite_backward_smooth_l1(const nsr_half *mlp_out, uint32_t stride, float density_bias,
                                                const float *t_starts, const float *t_ends, const nsr_half *rgb,
                                                uint32_t rgb_stride, const int32_t *packed_info,
                                                const float *background, const float *weights, const float *trans,
                                                const float *comp_rgb, const float *opacity, const float *gt_rgb,
                                                const float *acc2, float grad_scale, float *grad_rgb, float *grad_logit,
                                                uint32_t n_rays, void *stream)
{
    if (n_rays == 0) return NSR_OK;
    NSR_REQUIRE(packed_info && background && weights && trans && comp_rgb && opacity && gt_rgb && acc2 && grad_rgb &&
                    grad_logit, "nsr_composite_backward_smooth_l1: NULL pointer");
    hipLaunchKernelGGL(k_composite_backward, RAY_GRID(n_rays), (const __half *)mlp_out, stride, density_bias, t_starts,
                       t_ends, (const __half *)rgb, rgb_stride, packed_info, background, weights, trans, nullptr, nullptr,
                       nullptr, grad_rgb, grad_logit, n_rays, comp_rgb, opacity, gt_rgb, acc2, grad_scale, nullptr, nullptr,
                       nullptr);
    NSR_CHECK_LAUNCH("nsr_composite_backward_smooth_l1");
    return NSR_OK;
}

// The same pair with the loss reduction folded in: the forward leaves one (loss sum, valid rays) partial per block, every
// block of the backward sums them -- the one-workgroup reduction kernel between the two (10 us + a launch gap on the
// step's critical path) is gone.  partials: nsr_composite_l1_partials_floats(n_rays) floats, no initialisation needed.
extern "C" uint64_t nsr_composite_l1_partials_floats(uint32_t n_rays)
{
    return 2ull * ((n_rays + RAYS_PER_BLOCK - 1) / RAYS_PER_BLOCK);
}

extern "C" int nsr_composite_forward_smooth_l1(const nsr_half *mlp_out, uint32_t stride, float density_bias,
                                               const float *t_starts, const float *t_ends, const nsr_half *rgb,
                                               uint32_t rgb_stride, const int32_t *packed_info, const float *background,
                                               float *weights, float *trans, float *comp_rgb, float *opacity, float *depth,
                                               const float *gt_rgb, float *partials, uint32_t n_rays, void *stream)
{
    if (n_rays == 0) return NSR_OK;
    NSR_REQUIRE(packed_info && background && comp_rgb && opacity && depth && gt_rgb && partials,
                "nsr_composite_forward_smooth_l1: NULL pointer");
    hipLaunchKernelGGL(k_composite_forward, RAY_GRID(n_rays), (const __half *)mlp_out, stride, density_bias, t_starts,
                       t_ends, (const __half *)rgb, rgb_stride, packed_info, background, weights, trans, comp_rgb,
                       opacity, depth, n_rays, gt_rgb, partials);
    NSR_CHECK_LAUNCH("nsr_composite_forward_smooth_l1");
    return NSR_OK;
}

extern "C" int nsr_composite_backward_smooth_l1_partials(const nsr_half *mlp_out, uint32_t stride, float density_bias,
                                                         const float *t_starts, const float *t_ends, const nsr_half *rgb,
                                                         uint32_t rgb_stride, const int32_t *packed_info,
                                                         const float *background, const float *weights, const float *trans,
                                                         const float *comp_rgb, const float *opacity, const float *gt_rgb,
                                                         const float *partials, float *acc2, float grad_scale,
                                                         float *grad_rgb, float *grad_logit, uint32_t n_rays, void *stream)
{
    if (n_rays == 0) return NSR_OK;
    NSR_REQUIRE(packed_info && background && weights && trans && comp_rgb && opacity && gt_rgb && partials && acc2 &&
                    grad_rgb && grad_logit, "nsr_composite_backward_smooth_l1_partials: NULL pointer");
    hipLaunchKernelGGL(k_composite_backward, RAY_GRID(n_rays), (const __half *)mlp_out, stride, density_bias, t_starts,
                       t_ends, (const __half *)rgb, rgb_stride, packed_info, background, weights, trans, nullptr, nullptr,
                       nullptr, grad_rgb, grad_logit, n_rays, comp_rgb, opacity, gt_rgb, nullptr, grad_scale, nullptr,
                       partials, acc2);
    NSR_CHECK_LAUNCH("nsr_composite_backward_smooth_l1_partials");
    return NSR_OK;
}

// sample-partitioned forms of the compositing pair (k_composite_*_samples): one lane per kept sample.  ray_indices[n_samples]
// (int64, as the kept-row copy writes them) names each sample's ray, packed_info its segment; n_samples is the capacity of the
// sample arrays, n_samples_dev (may be NULL) the live count.  partials (may be NULL): nsr_composite_l1_partials_floats(n_rays)
// floats; forward and backward of a step take the same (n_rays, n_samples).
static uint32_t sp_grid(uint32_t n_rays, uint32_t n_samples)
{
    const uint32_t want = nsr_div_up(n_samples > n_rays ? n_samples : n_rays, SP_BLOCK);
    const uint32_t slots = (n_rays + RAYS_PER_BLOCK - 1) / RAYS_PER_BLOCK;  // loss-partial slots (one per block)
    const uint32_t g = want < slots ? want : slots;
    return g ? g : 1u;
}

extern "C" int nsr_composite_forward_samples(const nsr_half *mlp_out, uint32_t stride, float density_bias,
                                             const float *t_starts, const float *t_ends, const nsr_half *rgb,
                                             uint32_t rgb_stride, const int32_t *packed_info, const int64_t *ray_indices,
                                             const float *background, float *weights, float *trans, float *comp_rgb,
                                             float *opacity, float *depth, const float *gt_rgb, float *partials,
                                             uint32_t n_rays, uint32_t n_samples, const int32_t *n_samples_dev, void *stream)
{
    if (n_rays == 0) return NSR_OK;
    NSR_REQUIRE(packed_info && background && comp_rgb && opacity && depth && (n_samples == 0 || (weights && trans && ray_indices)),
                "nsr_composite_forward_samples: NULL pointer");
    NSR_REQUIRE(!partials || gt_rgb, "nsr_composite_forward_samples: the loss partials need gt_rgb");
    hipLaunchKernelGGL(k_composite_forward_samples, dim3(sp_grid(n_rays, n_samples)), dim3(SP_BLOCK), 0, (hipStream_t)stream,
                       (const __half *)mlp_out, stride, density_bias, t_starts, t_ends, (const __half *)rgb, rgb_stride,
                       packed_info, ray_indices, background, weights, trans, comp_rgb, opacity, depth, n_rays, n_samples,
                       n_samples_dev, gt_rgb, partials);
    NSR_CHECK_LAUNCH("nsr_composite_forward_samples");
    return NSR_OK;
}

// upstream gradients: either (grad_comp_rgb [+ grad_opacity, grad_depth, grad_weights]) or the masked smooth-L1 loss on
// (comp_rgb, opacity, gt_rgb) with its (sum, valid) either in acc2 (partials == NULL) or as the forward's partials (then
// acc2 receives the totals)
extern "C" int nsr_composite_backward_samples(const nsr_half *mlp_out, uint32_t stride, float density_bias,
                                              const float *t_starts, const float *t_ends, const nsr_half *rgb,
                                              uint32_t rgb_stride, const int32_t *packed_info, const int64_t *ray_indices,
                                              const float *background, const float *weights, const float *trans,
                                              const float *grad_comp_rgb, const float *grad_opacity, const float *grad_depth,
                                              const float *grad_weights, const float *comp_rgb, const float *opacity,
                                              const float *gt_rgb, const float *partials, float *acc2, float grad_scale,
                                              float *grad_rgb, float *grad_logit, uint32_t n_rays, uint32_t n_samples,
                                              const int32_t *n_samples_dev, void *stream)
{
    if (n_rays == 0 || n_samples == 0) return NSR_OK;
    NSR_REQUIRE(packed_info && ray_indices && background && weights && trans && grad_rgb && grad_logit,
                "nsr_composite_backward_samples: NULL pointer");
    NSR_REQUIRE((grad_comp_rgb != nullptr) != (comp_rgb != nullptr), "nsr_composite_backward_samples: either upstream "
                "gradients or the built-in loss");
    NSR_REQUIRE(!comp_rgb || (opacity && gt_rgb && acc2), "nsr_composite_backward_samples: the built-in loss needs opacity, "
                "gt_rgb, acc2");
    const uint32_t grid = sp_grid(n_rays, n_samples);
    hipLaunchKernelGGL(k_composite_backward_samples, dim3(grid), dim3(SP_BLOCK), 0, (hipStream_t)stream,
                       (const __half *)mlp_out, stride, density_bias, t_starts, t_ends, (const __half *)rgb, rgb_stride,
                       packed_info, ray_indices, background, weights, trans, grad_comp_rgb, grad_opacity, grad_depth, grad_rgb,
                       grad_logit, n_rays, n_samples, n_samples_dev, comp_rgb, opacity, gt_rgb, partials ? nullptr : acc2,
                       grad_scale, grad_weights, partials, grid, partials ? acc2 : nullptr);
    NSR_CHECK_LAUNCH("nsr_composite_backward_samples");
    return NSR_OK;
}

extern "C" int nsr_gather_train_rays(const float *images, const float *masks, const float *directions, const float *c2w,
                                     const int64_t *index, const int64_t *px, const int64_t *py, const float *background,
                                     int height, int width, int apply_mask, float *rays, float *rgb, float *fg,
                                     uint32_t n, void *stream)
{
    if (n == 0) return NSR_OK;
    NSR_REQUIRE(images && masks && directions && c2w && index && px && py && background && rays && rgb && fg,
                "nsr_gather_train_rays: NULL pointer");
    hipLaunchKernelGGL(k_gather_train_rays, EW_GRID(n), images, masks, directions, c2w, index, px, py, background,
                       height, width, apply_mask, rays, rgb, fg, n);
    NSR_CHECK_LAUNCH("nsr_gather_train_rays");
    return NSR_OK;
}

NSR_INTERNAL int nsr_copy_ray_prefix_rows_ex(const int32_t *packed_old, const int32_t *packed_new, uint32_t n_arrays,
                                           const void *const *src, void *const *dst, const uint32_t *row_bytes,
                                           const uint32_t *planes, const uint64_t *src_plane_bytes,
                                           const uint64_t *dst_plane_bytes, const float *rays_d, float *dirs_out,
                                           int64_t *ray_indices_out, const nsr_half *tex_src, uint32_t tex_src_stride,
                                           nsr_half *tex_in, uint32_t n_rays, void *stream)
{
    if (n_rays == 0) return NSR_OK;
    NSR_REQUIRE(packed_old && packed_new, "nsr_copy_ray_prefix_rows: NULL packed_info");
    NSR_REQUIRE(n_arrays <= 8, "nsr_copy_ray_prefix_rows: at most 8 arrays");
    NSR_REQUIRE(!dirs_out || rays_d, "nsr_copy_ray_prefix_rows: dirs_out needs rays_d");
    NSR_REQUIRE(!tex_in || (tex_src && rays_d && tex_src_stride >= 16 && (tex_src_stride & 7u) == 0),
                "nsr_copy_ray_prefix_rows: tex_in needs rays_d and 16-B aligned feature rows of >= 16 halfs");
    RowCopies rc;
    rc.n = n_arrays;
    for (uint32_t q = 0; q < n_arrays; ++q) {
        NSR_REQUIRE(src[q] && dst[q] && row_bytes[q] % 4 == 0, "nsr_copy_ray_prefix_rows: array %u: NULL or row not a multiple of 4 B", q);
        rc.a[q].src = (const uint32_t *)src[q];
        rc.a[q].dst = (uint32_t *)dst[q];
        rc.a[q].row_dwords = row_bytes[q] / 4;
        rc.a[q].planes = planes ? planes[q] : 1;
        rc.a[q].src_plane = planes ? src_plane_bytes[q] / 4 : 0;
        rc.a[q].dst_plane = planes ? dst_plane_bytes[q] / 4 : 0;
    }
    hipLaunchKernelGGL(k_copy_ray_prefix_rows, RAY_GRID(n_rays), packed_old, packed_new, rc, rays_d, dirs_out,
                       ray_indices_out, (const __half *)tex_src, tex_src_stride, (__half *)tex_in, n_rays);
    NSR_CHECK_LAUNCH("nsr_copy_ray_prefix_rows");
    return NSR_OK;
}

static int copy_kept_rows_impl(const int32_t *packed_marched, int32_t *packed_kept, const float *t_starts,
                               const float *t_ends, const float *x01, const nsr_half *enc, const nsr_half *out1,
                               const nsr_half *acts1, float *t_starts_out, float *t_ends_out, float *x01_out,
                               nsr_half *enc_out, nsr_half *out1_out, nsr_half *acts1_out, uint32_t n_levels,
                               uint32_t n_hidden, uint32_t marched_capacity, uint32_t kept_capacity, const float *rays_d,
                               int64_t *ray_indices_out, nsr_half *tex_in, uint32_t n_rays, const KeptScan *scan,
                               void *stream)
{
    if (n_rays == 0) return NSR_OK;
    NSR_REQUIRE(packed_marched && packed_kept && t_starts && t_ends && x01 && enc && out1 && acts1 && t_starts_out &&
                    t_ends_out && x01_out && enc_out && out1_out && acts1_out && rays_d && ray_indices_out && tex_in,
                "nsr_nerf_copy_kept_rows: NULL pointer");
    NSR_REQUIRE(n_hidden >= 1 && n_hidden <= 2, "nsr_nerf_copy_kept_rows: 1 or 2 hidden layers");
    KeptRows kr;
    kr.t0_s = t_starts; kr.t1_s = t_ends; kr.x01_s = x01;
    kr.enc_s = (const uint32_t *)enc; kr.out1_s = (const u32x4 *)out1; kr.acts_s = (const u32x4 *)acts1;
    kr.t0_d = t_starts_out; kr.t1_d = t_ends_out; kr.x01_d = x01_out;
    kr.enc_d = (uint32_t *)enc_out; kr.out1_d = (u32x4 *)out1_out; kr.acts_d = (u32x4 *)acts1_out;
    kr.enc_sp = marched_capacity; kr.enc_dp = kept_capacity;
    kr.acts_sl = (uint64_t)marched_capacity * 8; kr.acts_dl = (uint64_t)kept_capacity * 8;
    kr.planes = n_levels;
    KeptScan ks;
    memset(&ks, 0, sizeof(ks));
    if (scan) ks = *scan;
#define NSR_COPY(NH, P16)                                                                                                 \
    do {                                                                                                                  \
        if (scan)                                                                                                         \
            NSR_LAUNCH_STOP((k_copy_kept_rows<NH, P16, true>), dim3(nsr_div_up(n_rays, RAYS_PER_BLOCK)), dim3(R_BLOCK), 0,   \
                            (hipStream_t)stream, packed_marched, packed_kept, kr,                                        \
                            rays_d, ray_indices_out, (__half *)tex_in, n_rays, ks);                                       \
        else                                                                                                              \
            NSR_LAUNCH_STOP((k_copy_kept_rows<NH, P16, false>), dim3(nsr_div_up(n_rays, RAYS_PER_BLOCK)), dim3(R_BLOCK), 0,  \
                            (hipStream_t)stream, packed_marched, packed_kept, kr,                                        \
                            rays_d, ray_indices_out, (__half *)tex_in, n_rays, ks);                                       \
    } while (0)
    if (n_hidden == 1 && n_levels == 16) NSR_COPY(1, true);
    else if (n_hidden == 2 && n_levels == 16) NSR_COPY(2, true);
    else if (n_hidden == 1) NSR_COPY(1, false);
    else NSR_COPY(2, false);
#undef NSR_COPY
    NSR_CHECK_LAUNCH("nsr_nerf_copy_kept_rows");
    return NSR_OK;
}

extern "C" int nsr_nerf_copy_kept_rows(const int32_t *packed_marched, const int32_t *packed_kept, const float *t_starts,
                                       const float *t_ends, const float *x01, const nsr_half *enc, const nsr_half *out1,
                                       const nsr_half *acts1, float *t_starts_out, float *t_ends_out, float *x01_out,
                                       nsr_half *enc_out, nsr_half *out1_out, nsr_half *acts1_out, uint32_t n_levels,
                                       uint32_t n_hidden, uint32_t marched_capacity, uint32_t kept_capacity,
                                       const float *rays_d, int64_t *ray_indices_out, nsr_half *tex_in, uint32_t n_rays,
                                       void *stream)
{
    return copy_kept_rows_impl(packed_marched, const_cast<int32_t *>(packed_kept), t_starts, t_ends, x01, enc, out1, acts1,
                               t_starts_out, t_ends_out, x01_out, enc_out, out1_out, acts1_out, n_levels, n_hidden,
                               marched_capacity, kept_capacity, rays_d, ray_indices_out, tex_in, n_rays, nullptr, stream);
}

// ... with the packing folded in: packed_kept [n_rays][2], total_kept[1] (and the statistics of nsr_pack_from_counts_capped,
// stats may be NULL) are WRITTEN here from kept_counts + the block sums of nsr_visibility_prefix_sums; rays past
// `kept_capacity` samples are truncated as there.  One launch instead of nsr_pack_from_counts_capped + nsr_nerf_copy_kept_rows.
extern "C" int nsr_nerf_copy_kept_rows_scan(const int32_t *packed_marched, const int32_t *kept_counts,
                                            const int32_t *block_sums, int32_t *packed_kept, int32_t *total_kept,
                                            int32_t *stats, const float *t_starts, const float *t_ends, const float *x01,
                                            const nsr_half *enc, const nsr_half *out1, const nsr_half *acts1,
                                            float *t_starts_out, float *t_ends_out, float *x01_out, nsr_half *enc_out,
                                            nsr_half *out1_out, nsr_half *acts1_out, uint32_t n_levels, uint32_t n_hidden,
                                            uint32_t marched_capacity, uint32_t kept_capacity, const float *rays_d,
                                            int64_t *ray_indices_out, nsr_half *tex_in, uint32_t n_rays, void *stream)
{
    NSR_REQUIRE(kept_counts && block_sums && total_kept, "nsr_nerf_copy_kept_rows_scan: NULL pointer");
    NSR_REQUIRE(((uintptr_t)block_sums & 15u) == 0 && (!stats || ((uintptr_t)stats & 7u) == 0),
                "nsr_nerf_copy_kept_rows_scan: block_sums must be 16-byte, stats 8-byte aligned");
    NSR_REQUIRE(kept_capacity < 0x7fffffffu, "nsr_nerf_copy_kept_rows_scan: capacity must fit int32");
    KeptScan ks;
    ks.kept = kept_counts; ks.block_sums = block_sums; ks.total = total_kept; ks.stats = stats; ks.capacity = kept_capacity;
    return copy_kept_rows_impl(packed_marched, packed_kept, t_starts, t_ends, x01, enc, out1, acts1, t_starts_out, t_ends_out,
                               x01_out, enc_out, out1_out, acts1_out, n_levels, n_hidden, marched_capacity, kept_capacity,
                               rays_d, ray_indices_out, tex_in, n_rays, &ks, stream);
}

// kept counts per ray (as nsr_visibility_prefix) + one sum per block of 8 rays: block_sums holds nsr_div_up(n_rays, 8) words
extern "C" int nsr_visibility_prefix_sums(const nsr_half *mlp_out, uint32_t stride, float density_bias, const float *t_starts,
                                          const float *t_ends, const int32_t *packed_info, float early_stop_eps,
                                          int32_t *kept_counts, int32_t *block_sums, uint32_t n_rays, void *stream)
{
    if (n_rays == 0) return NSR_OK;
    NSR_REQUIRE(packed_info && kept_counts && block_sums, "nsr_visibility_prefix_sums: NULL pointer");
    hipLaunchKernelGGL(k_visibility_prefix_sums, dim3(nsr_div_up(n_rays, VIS_RPB)), dim3(VIS_BLOCK), 0, (hipStream_t)stream,
                       (const __half *)mlp_out, stride, density_bias, t_starts, t_ends, packed_info, early_stop_eps,
                       kept_counts, block_sums, n_rays);
    NSR_CHECK_LAUNCH("nsr_visibility_prefix_sums");
    return NSR_OK;
}

extern "C" int nsr_copy_ray_prefix_rows(const int32_t *packed_old, const int32_t *packed_new, uint32_t n_arrays,
                                        const void *const *src, void *const *dst, const uint32_t *row_bytes,
                                        const float *rays_d, float *dirs_out, int64_t *ray_indices_out, uint32_t n_rays,
                                        void *stream)
{
    return nsr_copy_ray_prefix_rows_ex(packed_old, packed_new, n_arrays, src, dst, row_bytes, nullptr, nullptr, nullptr,
                                       rays_d, dirs_out, ray_indices_out, nullptr, 0, nullptr, n_rays, stream);
}

extern "C" int nsr_prepare_train_rays(const float *images, const float *masks, const float *directions, const float *c2w,
                                      const float *u01, const float *background, int n_images, int height, int width,
                                      int apply_mask, const float *aabb, float jitter_step, float *rays, float *rays_o,
                                      float *rays_d, float *rgb, float *fg, float *t_min, float *t_max, uint32_t n,
                                      const int32_t *n_active, void *stream)
{
    if (n == 0) return NSR_OK;
    NSR_REQUIRE(images && masks && directions && c2w && u01 && background && aabb && rays && rays_o && rays_d && rgb &&
                    fg && t_min && t_max, "nsr_prepare_train_rays: NULL pointer");
    hipLaunchKernelGGL(k_prepare_train_rays, EW_GRID(n), images, masks, directions, c2w, u01, background, n_images,
                       height, width, apply_mask, aabb, jitter_step, rays, rays_o, rays_d, rgb, fg, t_min, t_max, n,
                       n_active);
    NSR_CHECK_LAUNCH("nsr_prepare_train_rays");
    return NSR_OK;
}

extern "C" int nsr_update_ray_count(const int32_t *n_samples, int32_t *n_rays, int32_t target_samples, int32_t max_rays,
                                    int64_t *rays_accum, void *stream)
{
    NSR_REQUIRE(n_samples && n_rays, "nsr_update_ray_count: NULL pointer");
    NSR_REQUIRE(target_samples >= 0 && max_rays > 0, "nsr_update_ray_count: target_samples must be >= 0, max_rays > 0");
    hipLaunchKernelGGL(k_update_ray_count, dim3(1), dim3(64), 0, (hipStream_t)stream, n_samples, n_rays, target_samples,
                       max_rays, (long long *)rays_accum);
    NSR_CHECK_LAUNCH("nsr_update_ray_count");
    return NSR_OK;
}
