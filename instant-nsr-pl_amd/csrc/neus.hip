// Fused NeuS step glue (reference models/neus.py:205-287 `NeuSModel.forward_`, models/geometry.py:158-210 `VolumeSDF.forward`,
// models/neus.py:117-139 `get_alpha`, models/texture.py:23-30, loss terms of systems/neus.py:96-130): everything between
// the marcher, the hash grid, the fp32 SDF network (csrc/vmlp.hip) and the colour network that the reference issues as
// ~200 elementwise / reduction launches through autograd.
//
//   k_neus_points        sample positions in unit coordinates (+ the six +-eps finite-difference taps, clamped to the box and
//                        scaled with the PLAIN AABB rule exactly as models/geometry.py:193-194), view directions
//   k_neus_shade_fwd     sdf gradient (analytic: (2 g_xyz + J^T g_enc) / 2r; finite differences: central differences + the
//                        7-point laplace), normal, SDF->alpha, colour-network input [feature | SH4(dir) | normal]; the
//                        per-sample loss sums (eikonal, sparsity, curvature) leave through one atomic per wave
//   k_neus_composite_fwd wave per ray: alpha compositing (weights, opacity, depth, colour, normal) + background
//   k_neus_loss_rays     one workgroup: L1 / MSE over valid rays, mask and opacity BCE (systems/criterions.py:155-159)
//   k_neus_composite_bwd wave per ray: loss gradient formed per ray, then d alpha / d colour logits per sample
//   k_neus_shade_bwd     d alpha, d colour input, eikonal / sparsity / curvature -> d (SDF network output) and either
//                        dL/d(sdf gradient) (analytic: seeds the double backward) or d (tap sdf) (finite differences)
#include "nsr_common.h"
#include <string.h>

namespace {

constexpr int EW_BLOCK = 256;
constexpr int R_BLOCK = 256;
constexpr int RAYS_PER_BLOCK = R_BLOCK / NSR_WAVE;
#define EW_GRID(n) dim3(nsr_div_up((n), EW_BLOCK)), dim3(EW_BLOCK), 0, (hipStream_t)stream
#define EW_GRID_CAPPED(n) dim3(nsr_div_up((n), EW_BLOCK) < 2048u ? nsr_div_up((n), EW_BLOCK) : 2048u), dim3(EW_BLOCK), 0, (hipStream_t)stream
#define RAY_GRID(n) dim3(nsr_div_up((n), RAYS_PER_BLOCK)), dim3(R_BLOCK), 0, (hipStream_t)stream

__device__ __forceinline__ float sigmoidf(float v) { return 1.f / (1.f + expf(-v)); }

// accumulator slots of the loss sums (float[16], zeroed by the caller before the forward pass)
enum { ACC_L1 = 0, ACC_MSE, ACC_VALID, ACC_MASK, ACC_OPAQUE, ACC_EIK, ACC_SPARSE, ACC_CURV, ACC_INV_S_GRAD, ACC_N };

__global__ void __launch_bounds__(EW_BLOCK)
k_neus_points(const float *__restrict__ rays_o, const float *__restrict__ rays_d, const int64_t *__restrict__ ri,
              const float *__restrict__ t0, const float *__restrict__ t1, float radius, float eps, int taps,
              float *__restrict__ x7 /* [1 + 6 taps][n][3] */, float *__restrict__ dirs, uint32_t n,
              const int32_t *__restrict__ n_dev)
{
    const uint32_t i = blockIdx.x * EW_BLOCK + threadIdx.x;
    if (i >= live_count(n, n_dev)) return;
    const int64_t r = ri[i];
    const float mid = (t0[i] + t1[i]) / 2.f;
    float p[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float d = rays_d[3 * r + k];
        p[k] = rays_o[3 * r + k] + d * mid;
        if (dirs) dirs[3ull * i + k] = d;
        x7[3ull * i + k] = (p[k] + radius) / (radius + radius);
    }
    if (taps) {
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const int axis = t >> 1;
            const float off = (t & 1) ? -eps : eps;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float q = k == axis ? p[k] + off : p[k] + 0.f;
                q = fminf(fmaxf(q, -radius), radius);
                x7[((uint64_t)(t + 1) * n + i) * 3 + k] = (q + radius) / (radius + radius);
            }
        }
    }
}

__device__ __forceinline__ void sh4(float x, float y, float z, float *o)
{
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
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
}

struct AlphaEval {
    float alpha, prev, next, num, den, ep, en, inv_s, dic_dtc, raw_a;
    bool s_live;
};
// models/neus.py:117-139
__device__ __forceinline__ AlphaEval neus_alpha(float sdf, const float *nrm, const float *d, float dist, float raw_inv_s,
                                                float anneal)
{
    AlphaEval e;
    e.inv_s = fminf(fmaxf(raw_inv_s, 1e-6f), 1e6f);
    e.s_live = raw_inv_s >= 1e-6f && raw_inv_s <= 1e6f;
    const float tc = d[0] * nrm[0] + d[1] * nrm[1] + d[2] * nrm[2];
    const float u1 = -tc * 0.5f + 0.5f, u2 = -tc;
    const float ic = -(fmaxf(u1, 0.f) * (1.f - anneal) + fmaxf(u2, 0.f) * anneal);
    e.dic_dtc = -((u1 > 0.f ? -0.5f : 0.f) * (1.f - anneal) + (u2 > 0.f ? -1.f : 0.f) * anneal);
    const float h = ic * dist * 0.5f;
    e.ep = sdf - h;
    e.en = sdf + h;
    e.prev = sigmoidf(e.ep * e.inv_s);
    e.next = sigmoidf(e.en * e.inv_s);
    e.num = (e.prev - e.next) + 1e-5f;
    e.den = e.prev + 1e-5f;
    e.raw_a = e.num / e.den;
    e.alpha = fminf(fmaxf(e.raw_a, 0.f), 1.f);
    return e;
}

template <bool FD, bool TEX_F32>
__global__ void __launch_bounds__(EW_BLOCK)
k_neus_shade_fwd(const float *__restrict__ sdf_out /* [n][16]: SDF network output, col 0 = sdf */,
                 const float *__restrict__ g_in /* analytic: [n][g_stride], cols 0..2 = d sdf / d(2x-1) */,
                 uint32_t g_stride, const float *__restrict__ dx01 /* analytic: J^T g_enc [n][3] */,
                 const float *__restrict__ tap_sdf /* FD: [6][n] */, float eps, float radius,
                 const float *__restrict__ dirs, const float *__restrict__ t0, const float *__restrict__ t1,
                 const float *__restrict__ inv_s_p, float anneal, uint32_t n_feat, float sparsity_scale,
                 float *__restrict__ grad /* [n][3] */, float *__restrict__ normal, float *__restrict__ alpha,
                 float *__restrict__ laplace, void *__restrict__ tex_in /* [n][32] half or float */,
                 float *__restrict__ acc, uint32_t n, const int32_t *__restrict__ n_dev)
{
    float s_eik = 0.f, s_sp = 0.f, s_curv = 0.f;
    const uint32_t n_live = live_count(n, n_dev);
    for (uint32_t i = blockIdx.x * EW_BLOCK + threadIdx.x; i < n_live; i += gridDim.x * EW_BLOCK) {
        const float sdf = sdf_out[16ull * i];
        float g[3], lap = 0.f;
        if (FD) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float a = tap_sdf[(uint64_t)(2 * k) * n + i], b = tap_sdf[(uint64_t)(2 * k + 1) * n + i];
                g[k] = 0.5f * (a - b) / eps;
                lap += a + b - 2.f * sdf;
            }
            lap = lap / (eps * eps);
            laplace[i] = lap;
            s_curv += fabsf(lap);
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k)
                g[k] = (2.f * g_in[(uint64_t)i * g_stride + k] + dx01[3ull * i + k]) / (radius + radius);
        }
        const float nrm2 = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
        const float inv = 1.f / fmaxf(nrm2, 1e-12f);  // F.normalize(p=2, eps=1e-12)
        float nv[3], dv[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            grad[3ull * i + k] = g[k];
            nv[k] = g[k] * inv;
            normal[3ull * i + k] = nv[k];
            dv[k] = dirs[3ull * i + k];
        }
        s_eik += (nrm2 - 1.f) * (nrm2 - 1.f);
        s_sp += expf(-sparsity_scale * fabsf(sdf));
        alpha[i] = neus_alpha(sdf, nv, dv, t1[i] - t0[i], inv_s_p[0], anneal).alpha;
        // colour-network input: [feature (n_feat) | SH4 of the direction (fp16-rounded, what tcnn hands back) | normal]
        float shv[16];
        {   // models/texture.py:24: dirs -> (d + 1) / 2, the encoder maps back 2u - 1
            const float ux = (dv[0] + 1.f) / 2.f, uy = (dv[1] + 1.f) / 2.f, uz = (dv[2] + 1.f) / 2.f;
            sh4(ux * 2.f - 1.f, uy * 2.f - 1.f, uz * 2.f - 1.f, shv);
        }
        if (TEX_F32) {
            float *row = reinterpret_cast<float *>(tex_in) + 32ull * i;
            for (uint32_t k = 0; k < n_feat; ++k) row[k] = sdf_out[16ull * i + k];
#pragma unroll
            for (int k = 0; k < 16; ++k) row[n_feat + k] = __half2float(__float2half_rn(shv[k]));
#pragma unroll
            for (int k = 0; k < 3; ++k) row[n_feat + 16 + k] = nv[k];
            for (uint32_t k = n_feat + 19; k < 32; ++k) row[k] = 0.f;
        } else {
            __half *row = reinterpret_cast<__half *>(tex_in) + 32ull * i;
            for (uint32_t k = 0; k < n_feat; ++k) row[k] = __float2half_rn(sdf_out[16ull * i + k]);
#pragma unroll
            for (int k = 0; k < 16; ++k) row[n_feat + k] = __float2half_rn(shv[k]);
#pragma unroll
            for (int k = 0; k < 3; ++k) row[n_feat + 16 + k] = __float2half_rn(nv[k]);
            for (uint32_t k = n_feat + 19; k < 32; ++k) row[k] = __float2half_rn(1.f);  // tcnn pads inputs with 1
        }
    }
    // one atomic per workgroup and quantity (same-address float atomics serialise: ~10 ns each)
    __shared__ float red[EW_BLOCK / 64][3];
    s_eik = wave_sum(s_eik);
    s_sp = wave_sum(s_sp);
    s_curv = FD ? wave_sum(s_curv) : 0.f;
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = s_eik; red[threadIdx.x >> 6][1] = s_sp; red[threadIdx.x >> 6][2] = s_curv; }
    __syncthreads();
    if (threadIdx.x < 3 && (FD || threadIdx.x < 2)) {
        float t = 0.f;
        for (int w = 0; w < EW_BLOCK / 64; ++w) t += red[w][threadIdx.x];
        unsafeAtomicAdd(acc + (threadIdx.x == 0 ? ACC_EIK : (threadIdx.x == 1 ? ACC_SPARSE : ACC_CURV)), t);
    }
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

template <bool RGB_F32>
__device__ __forceinline__ float load_rgb(const void *__restrict__ rgb, uint64_t i, int q)
{
    return RGB_F32 ? reinterpret_cast<const float *>(rgb)[16 * i + q]
                   : __half2float(reinterpret_cast<const __half *>(rgb)[16 * i + q]);
}

// models/neus.py:238-247,273-277 (render_weight_from_alpha, accumulate_along_rays x4, comp_rgb_full)
template <bool RGB_F32>
__global__ void __launch_bounds__(R_BLOCK)
k_neus_composite_fwd(const int32_t *__restrict__ packed, const float *__restrict__ alpha,
                     const void *__restrict__ rgb_raw /* [n][16]: colour logits (sigmoid applied here) */,
                     const float *__restrict__ normal, const float *__restrict__ t0, const float *__restrict__ t1,
                     const float *__restrict__ bg /* [3] (bg_stride 0) or per ray [n_rays][3] (bg_stride 3) */,
                     uint32_t bg_stride, float *__restrict__ weights, float *__restrict__ trans,
                     float *__restrict__ comp_rgb, float *__restrict__ opacity, float *__restrict__ depth,
                     float *__restrict__ comp_normal, float *__restrict__ comp_rgb_full, uint32_t n_rays)
{
    uint32_t r, start, count;
    if (!wave_ray(packed, n_rays, r, start, count)) return;
    const uint32_t lane = threadIdx.x & 63;
    float carry = 1.f;  // product of (1 - alpha) over the samples before this chunk
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // opacity, depth, rgb, normal
    for (uint32_t c = 0; c < count; c += 64) {
        const uint32_t k = c + lane;
        const bool ok = k < count;
        const uint64_t i = start + k;
        const float a = ok ? alpha[i] : 0.f;
        const float inc = wave_incl_scan_mul(1.f - a);
        float excl = __shfl_up(inc, 1, 64);
        if (lane == 0) excl = 1.f;
        const float T = carry * excl;
        const float w = T * a;
        if (ok) {
            weights[i] = w;
            trans[i] = T;
            acc[0] += w;
            acc[1] += w * ((t0[i] + t1[i]) / 2.f);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                acc[2 + q] += w * sigmoidf(load_rgb<RGB_F32>(rgb_raw, i, q));
                acc[5 + q] += w * normal[3 * i + q];
            }
        }
        carry *= __shfl(inc, 63, 64);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = wave_sum(acc[q]);
    if (lane == 0) {
        opacity[r] = acc[0];
        depth[r] = acc[1];
        const float nn = fmaxf(sqrtf(acc[5] * acc[5] + acc[6] * acc[6] + acc[7] * acc[7]), 1e-12f);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            comp_rgb[3ull * r + q] = acc[2 + q];
            comp_normal[3ull * r + q] = acc[5 + q] / nn;
            comp_rgb_full[3ull * r + q] = acc[2 + q] + bg[(uint64_t)r * bg_stride + q] * (1.f - acc[0]);
        }
    }
}

// systems/neus.py:96-117 per-ray terms; valid = rays_valid_full = opacity > 0 | opacity_bg > 0 (models/neus.py:283)
__global__ void __launch_bounds__(1024)
k_neus_loss_rays(const float *__restrict__ comp_rgb_full, const float *__restrict__ opacity,
                 const float *__restrict__ opacity_bg /* NULL: no learned background */, const float *__restrict__ gt,
                 const float *__restrict__ fg_mask, float *__restrict__ acc, uint32_t n_rays,
                 const int32_t *__restrict__ n_active)
{
    __shared__ float part[16][5];
    const uint32_t live = n_active ? (uint32_t)max(min(*n_active, (int32_t)n_rays), 0) : n_rays;
    float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};  // L1, MSE, valid, mask BCE, opaque BCE
    for (uint32_t r = threadIdx.x; r < live; r += 1024) {
        const float op = opacity[r];
        if (op > 0.f || (opacity_bg && opacity_bg[r] > 0.f)) {
            s[2] += 1.f;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const float d = comp_rgb_full[3ull * r + q] - gt[3ull * r + q];
                s[0] += fabsf(d);
                s[1] += d * d;
            }
        }
        const float o = fminf(fmaxf(op, 1e-3f), 1.f - 1e-3f);
        const float lo = logf(o), l1 = logf(1.f - o);
        const float m = fg_mask ? fg_mask[r] : 1.f;
        s[3] += -(m * lo + (1.f - m) * l1);
        s[4] += -(o * lo + (1.f - o) * l1);
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) s[q] = wave_sum(s[q]);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int q = 0; q < 5; ++q) part[threadIdx.x >> 6][q] = s[q];
    __syncthreads();
    if (threadIdx.x < 5) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += part[w][threadIdx.x];
        acc[threadIdx.x] = t;  // ACC_L1 .. ACC_OPAQUE
        if (threadIdx.x == 0) acc[ACC_N] = (float)live;
    }
}

struct NeusLossWeights {
    float rgb_l1, rgb_mse, mask, opaque, eikonal, sparsity, curvature, sparsity_scale;
};

// upstream gradients of a caller-owned loss (nsr.models.FusedNeuSModel: the reference's system forms its loss in torch on the
// model's output dict, systems/neus.py:96-139); every pointer may be NULL = zero.  With any of them set the built-in loss
// terms are off (their weights are passed as zero).
struct NeusUpstream {
    const float *comp_rgb_full, *comp_rgb, *opacity, *depth;  // per ray: [R][3], [R][3], [R], [R]
    const float *weights;                                     // per sample [n]
    const float *sdf, *sdf_grad, *laplace;                    // per sample [n], [n][3], [n]   (shade backward)
};

template <bool RGB_F32>
__global__ void __launch_bounds__(R_BLOCK)
k_neus_composite_bwd(const int32_t *__restrict__ packed, const float *__restrict__ alpha,
                     const void *__restrict__ rgb_raw, const float *__restrict__ weights, const float *__restrict__ trans,
                     const float *__restrict__ bg, uint32_t bg_stride, const float *__restrict__ opacity_bg,
                     const float *__restrict__ comp_rgb_full, const float *__restrict__ opacity,
                     const float *__restrict__ gt, const float *__restrict__ fg_mask, const float *__restrict__ acc,
                     NeusLossWeights lw, float loss_scale, float *__restrict__ d_alpha,
                     float *__restrict__ d_rgb_raw /* [n][16] fp32, cols 0..2 (3..15 zeroed) */,
                     float *__restrict__ d_bg /* per-ray background: dL/d comp_rgb_bg [n_rays][3] (may be NULL) */,
                     uint32_t n_rays, const int32_t *__restrict__ n_active, const NeusUpstream up,
                     const float *__restrict__ t0, const float *__restrict__ t1,
                     int32_t *__restrict__ guard /* overflow guard of a trainer (nsr_common.h NsrGuard) or NULL */, int parity)
{
    // the first kernel of a step's backward: the OTHER step parity's found-inf flag (the previous step's, read by its
    // optimizer kernels long ago) is cleared for the next step; a non-finite loss gradient raises this step's
    if (guard && blockIdx.x == 0 && threadIdx.x == 0) guard[1 - parity] = 0;
    uint32_t r, start, count;
    if (!wave_ray(packed, n_rays, r, start, count)) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t live = n_active ? (uint32_t)max(min(*n_active, (int32_t)n_rays), 0) : n_rays;
    const float op = opacity[r];
    const float n_valid = fmaxf(acc[ACC_VALID], 1.f), n_r = fmaxf(acc[ACC_N], 1.f);
    float dC[3] = {0.f, 0.f, 0.f}, dO = 0.f;
    // the caller's loss: dL/d comp_rgb_full (-> colours and, through the background blend, opacity), dL/d comp_rgb (colours
    // only), dL/d opacity, dL/d depth, dL/d weights
    float dCr[3] = {0.f, 0.f, 0.f}, dD = 0.f;
    if (up.comp_rgb_full)
#pragma unroll
        for (int q = 0; q < 3; ++q) dC[q] = up.comp_rgb_full[3ull * r + q];
    if (up.comp_rgb)
#pragma unroll
        for (int q = 0; q < 3; ++q) dCr[q] = up.comp_rgb[3ull * r + q];
    if (up.opacity) dO = up.opacity[r];
    if (up.depth) dD = up.depth[r];
    const bool external = up.comp_rgb_full || up.comp_rgb || up.opacity || up.depth || up.weights;
    if (external) {
#pragma unroll
        for (int q = 0; q < 3; ++q) dO -= bg[(uint64_t)r * bg_stride + q] * dC[q];
    } else if (r < live) {
        if (op > 0.f || (opacity_bg && opacity_bg[r] > 0.f)) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const float d = comp_rgb_full[3ull * r + q] - gt[3ull * r + q];
                const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
                dC[q] = loss_scale * (lw.rgb_l1 * sg + lw.rgb_mse * 2.f * d) / (3.f * n_valid);
            }
        }
        if (op >= 1e-3f && op <= 1.f - 1e-3f) {  // torch.clamp passes the gradient inside [min, max]
            const float m = fg_mask ? fg_mask[r] : 1.f;
            dO += loss_scale * lw.mask * (-(m / op) + (1.f - m) / (1.f - op)) / n_r;
            dO += loss_scale * lw.opaque * (-(logf(op) - logf(1.f - op))) / n_r;
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) dO -= bg[(uint64_t)r * bg_stride + q] * dC[q];  // comp_rgb_full = comp_rgb + bg (1 - opacity)
    }
    if (d_bg && (threadIdx.x & 63) == 0)
#pragma unroll
        for (int q = 0; q < 3; ++q) d_bg[3ull * r + q] = dC[q] * (1.f - op);
    bool bad = !(fabsf(dC[0]) + fabsf(dC[1]) + fabsf(dC[2]) + fabsf(dCr[0]) + fabsf(dCr[1]) + fabsf(dCr[2]) + fabsf(dO) +
                 fabsf(dD) <= 3.4028235e38f);
    // per sample: g_w = dC . rgb + dO ; d alpha_i = g_w_i T_i - (sum_{j>i} g_w_j w_j) / max(1 - alpha_i, 1e-10)
    float carry = 0.f;  // sum of g_w_j w_j over the samples AFTER this chunk (walking backwards)
    for (uint32_t c = 0; c < count; c += 64) {
        const uint32_t k = c + lane;
        const bool ok = k < count;
        const uint64_t i = (uint64_t)start + count - 1 - k;
        float v = 0.f, gw = 0.f, w = 0.f, a = 0.f, rgb[3] = {0.f, 0.f, 0.f};
        if (ok) {
            w = weights[i];
            a = alpha[i];
#pragma unroll
            for (int q = 0; q < 3; ++q) rgb[q] = sigmoidf(load_rgb<RGB_F32>(rgb_raw, i, q));
            gw = (dC[0] + dCr[0]) * rgb[0] + (dC[1] + dCr[1]) * rgb[1] + (dC[2] + dCr[2]) * rgb[2] + dO;
            if (dD != 0.f) gw += dD * ((t0[i] + t1[i]) * 0.5f);
            if (up.weights) gw += up.weights[i];
            v = gw * w;
        }
        const float inc = wave_incl_scan_add(v);
        if (ok) {
            const float after = carry + (inc - v);
            const float da = gw * trans[i] - after / fmaxf(1.f - a, 1e-10f);
            d_alpha[i] = da;
            bad |= !(fabsf(da) <= 3.4028235e38f);
            float *row = d_rgb_raw + 16 * i;
#pragma unroll
            for (int q = 0; q < 3; ++q) row[q] = w * (dC[q] + dCr[q]) * rgb[q] * (1.f - rgb[q]);
#pragma unroll
            for (int q = 3; q < 16; ++q) row[q] = 0.f;
        }
        carry += __shfl(inc, 63, 64);
    }
    if (guard && __any(bad) && lane == 0) atomicOr(guard + parity, 1);
}

template <bool FD>
__global__ void __launch_bounds__(EW_BLOCK)
k_neus_shade_bwd(const float *__restrict__ sdf_out, const float *__restrict__ grad, const float *__restrict__ normal,
                 const float *__restrict__ dirs, const float *__restrict__ t0, const float *__restrict__ t1,
                 const float *__restrict__ inv_s_p, float anneal, const float *__restrict__ laplace, float eps,
                 float radius, const float *__restrict__ d_alpha, const float *__restrict__ d_tex_in /* [n][32] */,
                 uint32_t n_feat, NeusLossWeights lw, float loss_scale, float n_samples,
                 float *__restrict__ d_out /* [n][16] */, float *__restrict__ gx /* analytic: dL/d(dx01) [n][3] */,
                 float *__restrict__ p_in /* analytic: [n][p_stride], cols 0..2 written */, uint32_t p_stride,
                 float *__restrict__ d_taps /* FD: [6][n] */, float *__restrict__ acc, uint32_t n,
                 const int32_t *__restrict__ n_dev, const NeusUpstream up, int32_t *__restrict__ guard, int parity)
{
    float gs_local = 0.f;
    bool bad = false;  // a non-finite gradient reached the SDF network's output (e.g. the fp16 colour network overflowed)
    const uint32_t n_live = live_count(n, n_dev);
    for (uint32_t i = blockIdx.x * EW_BLOCK + threadIdx.x; i < n_live; i += gridDim.x * EW_BLOCK) {
        const float sdf = sdf_out[16ull * i];
        float g[3], nv[3], dv[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { g[k] = grad[3ull * i + k]; nv[k] = normal[3ull * i + k]; dv[k] = dirs[3ull * i + k]; }
        const float dist = t1[i] - t0[i];
        const AlphaEval e = neus_alpha(sdf, nv, dv, dist, inv_s_p[0], anneal);
        float ga = d_alpha[i];
        if (!(e.raw_a >= 0.f && e.raw_a <= 1.f)) ga = 0.f;  // clip(0, 1)
        const float da_dprev = (e.den - e.num) / (e.den * e.den), da_dnext = -1.f / e.den;
        const float gp = ga * da_dprev * e.prev * (1.f - e.prev), gn = ga * da_dnext * e.next * (1.f - e.next);
        float d_sdf = (gp + gn) * e.inv_s;
        const float g_h = (-gp + gn) * e.inv_s;
        const float g_tc = g_h * dist * 0.5f * e.dic_dtc;
        gs_local += e.s_live ? (gp * e.ep + gn * e.en) : 0.f;
        // d normal: from alpha (through the cosine) and from the colour network's input columns
        const float *dt = d_tex_in + 32ull * i;
        float dn[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) dn[k] = g_tc * dv[k] + dt[n_feat + 16 + k];
        // normal = g / max(|g|, 1e-12)
        const float nrm2 = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
        float G[3];
        if (nrm2 > 1e-12f) {
            const float dot = nv[0] * dn[0] + nv[1] * dn[1] + nv[2] * dn[2];
#pragma unroll
            for (int k = 0; k < 3; ++k) G[k] = (dn[k] - nv[k] * dot) / nrm2;
            const float ce = loss_scale * lw.eikonal * 2.f * (nrm2 - 1.f) / n_samples;  // d mean((|g| - 1)^2) / d g
#pragma unroll
            for (int k = 0; k < 3; ++k) G[k] += ce * nv[k];
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) G[k] = dn[k] * 1e12f;
        }
        if (up.sdf_grad)  // the caller's loss on sdf_grad_samples (the eikonal term of systems/neus.py:106 formed outside)
#pragma unroll
            for (int k = 0; k < 3; ++k) G[k] += up.sdf_grad[3ull * i + k];
        if (up.sdf) d_sdf += up.sdf[i];
        // sparsity: mean(exp(-scale |sdf|))
        if (lw.sparsity != 0.f) {
            const float sg = sdf > 0.f ? 1.f : (sdf < 0.f ? -1.f : 0.f);
            d_sdf += loss_scale * lw.sparsity * (-lw.sparsity_scale * sg) * expf(-lw.sparsity_scale * fabsf(sdf)) / n_samples;
        }
        float *row = d_out + 16ull * i;
        if (FD) {
            const float lap = laplace[i];
            float cl = lw.curvature != 0.f
                ? loss_scale * lw.curvature * (lap > 0.f ? 1.f : (lap < 0.f ? -1.f : 0.f)) / (n_samples * eps * eps) : 0.f;
            if (up.laplace) cl += up.laplace[i] / (eps * eps);  // laplace = sum (f+ + f- - 2 f) / eps^2
            d_sdf += -6.f * cl;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float h = 0.5f * G[k] / eps;
                d_taps[(uint64_t)(2 * k) * n + i] = h + cl;
                d_taps[(uint64_t)(2 * k + 1) * n + i] = -h + cl;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                gx[3ull * i + k] = G[k] / (radius + radius);     // d grad / d (J^T g_enc)
                p_in[(uint64_t)i * p_stride + k] = G[k] / radius;  // d grad / d g_xyz = 2 / (2 r)
            }
        }
        row[0] = d_sdf + dt[0];
        for (uint32_t k = 1; k < n_feat; ++k) row[k] = dt[k];
        for (uint32_t k = n_feat; k < 16; ++k) row[k] = 0.f;
        // (column 0 and the normal's three columns of the colour network's input gradient stand for all of them: an overflow in
        // its hidden layers reaches every input column through the dense first layer; checking each of the n_feat copied
        // columns cost this kernel 26-34 us)
        bad |= !(fabsf(d_sdf + dt[0]) + fabsf(G[0]) + fabsf(G[1]) + fabsf(G[2]) <= 3.4028235e38f);
    }
    if (guard && __any(bad) && (threadIdx.x & 63) == 0) atomicOr(guard + parity, 1);
    __shared__ float red[EW_BLOCK / 64];
    gs_local = wave_sum(gs_local);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = gs_local;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < EW_BLOCK / 64; ++w) t += red[w];
        if (t != 0.f) unsafeAtomicAdd(acc + ACC_INV_S_GRAD, t);
    }
}

// ---- NeRF++ background branch (models/neus.py:169-203 `forward_bg_`): VolumeDensity on the contracted space with an fp32
// VanillaMLP head (models/geometry.py:116-130), trunc_exp density, VolumeRadiance colour head, density compositing ------

// kept[r] = #{ i in ray r : T_i >= eps } from fp32 logits (ray_marching's sigma_fn pruning, alpha_thre == 0)
__global__ void __launch_bounds__(R_BLOCK)
k_bg_visibility(const float *__restrict__ out16, float bias, const float *__restrict__ t0, const float *__restrict__ t1,
                const int32_t *__restrict__ packed, float eps, int32_t *__restrict__ kept, uint32_t n_rays)
{
    uint32_t r, start, count;
    if (!wave_ray(packed, n_rays, r, start, count)) return;
    const uint32_t lane = threadIdx.x & 63;
    float carry = 1.f;
    uint32_t n_kept = 0;
    for (uint32_t c = 0; c < count; c += 64) {
        const uint32_t k = c + lane;
        const bool ok = k < count;
        float keep = 1.f;
        if (ok) {
            const uint64_t i = (uint64_t)start + k;
            const float sigma = expf(out16[16 * i] + bias);
            keep = 1.f - (1.f - expf(-sigma * (t1[i] - t0[i])));
        }
        const float inc = wave_incl_scan_mul(keep);
        float exc = __shfl_up(inc, 1, 64);
        if (lane == 0) exc = 1.f;
        n_kept += (uint32_t)__popcll(__ballot(ok && carry * exc >= eps));
        carry *= __shfl(inc, 63, 64);
        if (carry < eps) break;  // wave-uniform: everything after is invisible
    }
    if (lane == 0) kept[r] = (int32_t)n_kept;
}

// colour-head input of the background: [feature (n_feat, column 0 = the density logit) | SH4(dir), fp16-rounded]
__global__ void __launch_bounds__(EW_BLOCK)
k_bg_texture_input(const float *__restrict__ out16, uint32_t n_feat, const float *__restrict__ rays_d,
                   const int64_t *__restrict__ ri, float *__restrict__ tex_in, uint32_t stride, uint32_t n,
                   const int32_t *__restrict__ n_dev)
{
    const uint32_t i = blockIdx.x * EW_BLOCK + threadIdx.x;
    if (i >= live_count(n, n_dev)) return;
    const int64_t r = ri[i];
    float shv[16];
    {
        const float ux = (rays_d[3 * r] + 1.f) / 2.f, uy = (rays_d[3 * r + 1] + 1.f) / 2.f, uz = (rays_d[3 * r + 2] + 1.f) / 2.f;
        sh4(ux * 2.f - 1.f, uy * 2.f - 1.f, uz * 2.f - 1.f, shv);
    }
    float *row = tex_in + (uint64_t)i * stride;
    for (uint32_t k = 0; k < n_feat; ++k) row[k] = out16[16ull * i + k];
#pragma unroll
    for (int k = 0; k < 16; ++k) row[n_feat + k] = __half2float(__float2half_rn(shv[k]));
    for (uint32_t k = n_feat + 16; k < stride; ++k) row[k] = 0.f;
}

// render_weight_from_density + accumulate_along_rays (weights, opacity, depth, colour) + background colour
__global__ void __launch_bounds__(R_BLOCK)
k_bg_composite_fwd(const int32_t *__restrict__ packed, const float *__restrict__ out16, float bias,
                   const float *__restrict__ rgb_raw, const float *__restrict__ t0, const float *__restrict__ t1,
                   const float *__restrict__ bg, float *__restrict__ weights, float *__restrict__ trans,
                   float *__restrict__ comp_rgb, float *__restrict__ opacity, float *__restrict__ depth, uint32_t n_rays)
{
    uint32_t r, start, count;
    if (!wave_ray(packed, n_rays, r, start, count)) return;
    const uint32_t lane = threadIdx.x & 63;
    float carry = 1.f;
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};  // opacity, depth, rgb
    for (uint32_t c = 0; c < count; c += 64) {
        const uint32_t k = c + lane;
        const bool ok = k < count;
        const uint64_t i = (uint64_t)start + k;
        float a = 0.f;
        if (ok) a = 1.f - expf(-expf(out16[16 * i] + bias) * (t1[i] - t0[i]));
        const float inc = wave_incl_scan_mul(1.f - a);
        float excl = __shfl_up(inc, 1, 64);
        if (lane == 0) excl = 1.f;
        const float T = carry * excl;
        const float w = T * a;
        if (ok) {
            weights[i] = w;
            trans[i] = T;
            acc[0] += w;
            acc[1] += w * ((t0[i] + t1[i]) / 2.f);
#pragma unroll
            for (int q = 0; q < 3; ++q) acc[2 + q] += w * sigmoidf(rgb_raw[16 * i + q]);
        }
        carry *= __shfl(inc, 63, 64);
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) acc[q] = wave_sum(acc[q]);
    if (lane == 0) {
        opacity[r] = acc[0];
        depth[r] = acc[1];
#pragma unroll
        for (int q = 0; q < 3; ++q) comp_rgb[3ull * r + q] = acc[2 + q] + bg[q] * (1.f - acc[0]);
    }
}

// d comp_rgb_bg -> d density logit, d colour logits.  sigma = trunc_exp(logit + bias) (backward: g exp(min(x, 15)),
// models/utils.py:55-66); alpha = 1 - exp(-sigma dt):  d sigma_i = dt_i [ g_w_i T_i (1 - alpha_i) - sum_{j>i} g_w_j w_j ]
__global__ void __launch_bounds__(R_BLOCK)
k_bg_composite_bwd(const int32_t *__restrict__ packed, const float *__restrict__ out16, float bias,
                   const float *__restrict__ rgb_raw, const float *__restrict__ weights, const float *__restrict__ trans,
                   const float *__restrict__ t0, const float *__restrict__ t1, const float *__restrict__ bg,
                   const float *__restrict__ d_comp /* [n_rays][3] */, float *__restrict__ d_logit,
                   float *__restrict__ d_rgb_raw /* [n][16] */, uint32_t n_rays)
{
    uint32_t r, start, count;
    if (!wave_ray(packed, n_rays, r, start, count)) return;
    const uint32_t lane = threadIdx.x & 63;
    float dC[3], dO = 0.f;
#pragma unroll
    for (int q = 0; q < 3; ++q) { dC[q] = d_comp[3ull * r + q]; dO -= bg[q] * dC[q]; }
    float carry = 0.f;
    for (uint32_t c = 0; c < count; c += 64) {
        const uint32_t k = c + lane;
        const bool ok = k < count;
        const uint64_t i = (uint64_t)start + count - 1 - k;
        float v = 0.f, gw = 0.f, w = 0.f, rgb[3] = {0.f, 0.f, 0.f};
        if (ok) {
            w = weights[i];
#pragma unroll
            for (int q = 0; q < 3; ++q) rgb[q] = sigmoidf(rgb_raw[16 * i + q]);
            gw = dC[0] * rgb[0] + dC[1] * rgb[1] + dC[2] * rgb[2] + dO;
            v = gw * w;
        }
        const float inc = wave_incl_scan_add(v);
        if (ok) {
            const float after = carry + (inc - v);
            const float x = out16[16 * i] + bias, dt = t1[i] - t0[i];
            const float keep = expf(-expf(x) * dt);  // 1 - alpha
            d_logit[i] = dt * (gw * trans[i] * keep - after) * expf(fminf(x, 15.f));
            float *row = d_rgb_raw + 16 * i;
#pragma unroll
            for (int q = 0; q < 3; ++q) row[q] = w * dC[q] * rgb[q] * (1.f - rgb[q]);
#pragma unroll
            for (int q = 3; q < 16; ++q) row[q] = 0.f;
        }
        carry += __shfl(inc, 63, 64);
    }
}

// d (density network output) [n][16] = [d logit + d feature_0 | d feature_1.. | 0]
__global__ void __launch_bounds__(EW_BLOCK)
k_bg_join(const float *__restrict__ d_logit, const float *__restrict__ d_tex, uint32_t stride, uint32_t n_feat,
          float *__restrict__ d_out16, uint32_t n, const int32_t *__restrict__ n_dev)
{
    const uint32_t i = blockIdx.x * EW_BLOCK + threadIdx.x;
    if (i >= live_count(n, n_dev)) return;
    const float *dt = d_tex + (uint64_t)i * stride;
    float *row = d_out16 + 16ull * i;
    row[0] = d_logit[i] + dt[0];
    for (uint32_t k = 1; k < n_feat; ++k) row[k] = dt[k];
    for (uint32_t k = n_feat; k < 16; ++k) row[k] = 0.f;
}

}  // namespace

extern "C" int nsr_neus_points(const float *rays_o, const float *rays_d, const int64_t *ray_indices, const float *t_starts,
                               const float *t_ends, float radius, float eps, int taps, float *x7, float *dirs, uint32_t n,
                               const int32_t *n_dev, void *stream)
{
    if (n == 0) return NSR_OK;
    NSR_REQUIRE(rays_o && rays_d && ray_indices && t_starts && t_ends && x7, "nsr_neus_points: NULL pointer");
    NSR_REQUIRE(radius > 0.f && (!taps || eps > 0.f), "nsr_neus_points: radius / eps must be positive");
    hipLaunchKernelGGL(k_neus_points, EW_GRID(n), rays_o, rays_d, ray_indices, t_starts, t_ends, radius, eps, taps, x7,
                       dirs, n, n_dev);
    NSR_CHECK_LAUNCH("nsr_neus_points");
    return NSR_OK;
}

extern "C" int nsr_neus_shade_forward(const float *sdf_out, const float *g_in, uint32_t g_stride, const float *dx01,
                                      const float *tap_sdf, float eps, float radius, const float *dirs,
                                      const float *t_starts, const float *t_ends, const float *inv_s,
                                      float cos_anneal_ratio, uint32_t n_feat, float sparsity_scale, float *grad,
                                      float *normal, float *alpha, float *laplace, void *tex_in, int tex_is_f32,
                                      float *acc, uint32_t n, const int32_t *n_dev, void *stream)
{
    if (n == 0) return NSR_OK;
    const bool fd = tap_sdf != nullptr;
    NSR_REQUIRE(sdf_out && dirs && t_starts && t_ends && inv_s && grad && normal && alpha && tex_in && acc,
                "nsr_neus_shade_forward: NULL pointer");
    NSR_REQUIRE(fd ? (laplace && eps > 0.f) : (g_in && dx01 && g_stride >= 3),
                "nsr_neus_shade_forward: finite differences need tap_sdf/laplace/eps, analytic needs g_in/dx01");
    NSR_REQUIRE(n_feat >= 1 && n_feat + 19 <= 32, "nsr_neus_shade_forward: n_feat=%u unsupported", n_feat);
#define SHADE(FDV, F32V)                                                                                              \
    hipLaunchKernelGGL((k_neus_shade_fwd<FDV, F32V>), EW_GRID_CAPPED(n), sdf_out, g_in, g_stride, dx01, tap_sdf, eps, radius,  \
                       dirs, t_starts, t_ends, inv_s, cos_anneal_ratio, n_feat, sparsity_scale, grad, normal, alpha,   \
                       laplace, tex_in, acc, n, n_dev)
    if (fd) { if (tex_is_f32) SHADE(true, true); else SHADE(true, false); }
    else { if (tex_is_f32) SHADE(false, true); else SHADE(false, false); }
#undef SHADE
    NSR_CHECK_LAUNCH("nsr_neus_shade_forward");
    return NSR_OK;
}

extern "C" int nsr_neus_composite_forward(const int32_t *packed_info, const float *alpha, const void *rgb_raw,
                                          int rgb_is_f32, const float *normal, const float *t_starts, const float *t_ends,
                                          const float *background, uint32_t background_stride, float *weights,
                                          float *trans, float *comp_rgb, float *opacity, float *depth,
                                          float *comp_normal, float *comp_rgb_full, uint32_t n_rays, void *stream)
{
    if (n_rays == 0) return NSR_OK;
    NSR_REQUIRE(background_stride == 0 || background_stride == 3, "nsr_neus_composite_forward: background_stride is 0 or 3");
    NSR_REQUIRE(packed_info && background && weights && trans && comp_rgb && opacity && depth && comp_normal && comp_rgb_full,
                "nsr_neus_composite_forward: NULL pointer");
    if (rgb_is_f32)
        hipLaunchKernelGGL(k_neus_composite_fwd<true>, RAY_GRID(n_rays), packed_info, alpha, rgb_raw, normal, t_starts,
                           t_ends, background, background_stride, weights, trans, comp_rgb, opacity, depth, comp_normal,
                           comp_rgb_full, n_rays);
    else
        hipLaunchKernelGGL(k_neus_composite_fwd<false>, RAY_GRID(n_rays), packed_info, alpha, rgb_raw, normal, t_starts,
                           t_ends, background, background_stride, weights, trans, comp_rgb, opacity, depth, comp_normal,
                           comp_rgb_full, n_rays);
    NSR_CHECK_LAUNCH("nsr_neus_composite_forward");
    return NSR_OK;
}

extern "C" int nsr_neus_loss_rays(const float *comp_rgb_full, const float *opacity, const float *opacity_bg,
                                  const float *gt_rgb, const float *fg_mask, float *acc, uint32_t n_rays,
                                  const int32_t *n_active, void *stream)
{
    NSR_REQUIRE(acc && (n_rays == 0 || (comp_rgb_full && opacity && gt_rgb)), "nsr_neus_loss_rays: NULL pointer");
    hipLaunchKernelGGL(k_neus_loss_rays, dim3(1), dim3(1024), 0, (hipStream_t)stream, comp_rgb_full, opacity, opacity_bg,
                       gt_rgb, fg_mask, acc, n_rays, n_active);
    NSR_CHECK_LAUNCH("nsr_neus_loss_rays");
    return NSR_OK;
}

static NeusUpstream make_upstream(const NsrNeusUpstream *u)
{
    NeusUpstream up;
    memset(&up, 0, sizeof(up));
    if (u) {
        up.comp_rgb_full = u->comp_rgb_full; up.comp_rgb = u->comp_rgb; up.opacity = u->opacity; up.depth = u->depth;
        up.weights = u->weights; up.sdf = u->sdf_samples; up.sdf_grad = u->sdf_grad_samples; up.laplace = u->sdf_laplace_samples;
    }
    return up;
}

extern "C" int nsr_neus_composite_backward_ex(const int32_t *packed_info, const float *alpha, const void *rgb_raw,
                                              int rgb_is_f32, const float *weights, const float *trans,
                                              const float *background, uint32_t background_stride,
                                              const float *opacity_bg, const float *comp_rgb_full, const float *opacity,
                                              const float *gt_rgb, const float *fg_mask, const float *acc,
                                              const float *loss_weights8, float loss_scale, float *d_alpha,
                                              float *d_rgb_raw, float *d_background, uint32_t n_rays,
                                              const int32_t *n_active, const NsrNeusUpstream *upstream,
                                              const float *t_starts, const float *t_ends, void *stream)
{
    if (n_rays == 0) return NSR_OK;
    const NeusUpstream up = make_upstream(upstream);
    const bool external = up.comp_rgb_full || up.comp_rgb || up.opacity || up.depth || up.weights;
    NSR_REQUIRE(packed_info && background && weights && trans && opacity && acc && loss_weights8 && d_alpha && d_rgb_raw &&
                    (external || (comp_rgb_full && gt_rgb)), "nsr_neus_composite_backward: NULL pointer");
    NSR_REQUIRE(!up.depth || (t_starts && t_ends), "nsr_neus_composite_backward: an upstream depth gradient needs t_starts / t_ends");
    NSR_REQUIRE(background_stride == 0 || background_stride == 3, "nsr_neus_composite_backward: background_stride is 0 or 3");
    NeusLossWeights lw;
    memcpy(&lw, loss_weights8, sizeof(lw));
    // a trainer's overflow guard (nsr_overflow_guard): this is the first kernel of a step's backward -- the step takes the
    // other parity's flag (see the kernel); the shade backward, the table backwards and the optimizer kernels follow it
    if (nsr_guard.state) nsr_guard.parity ^= 1;
    if (rgb_is_f32)
        hipLaunchKernelGGL(k_neus_composite_bwd<true>, RAY_GRID(n_rays), packed_info, alpha, rgb_raw, weights, trans, background,
                           background_stride, opacity_bg, comp_rgb_full, opacity, gt_rgb, fg_mask, acc, lw, loss_scale,
                           d_alpha, d_rgb_raw, d_background, n_rays, n_active, up, t_starts, t_ends, nsr_guard.state,
                           nsr_guard.parity);
    else
        hipLaunchKernelGGL(k_neus_composite_bwd<false>, RAY_GRID(n_rays), packed_info, alpha, rgb_raw, weights, trans, background,
                           background_stride, opacity_bg, comp_rgb_full, opacity, gt_rgb, fg_mask, acc, lw, loss_scale,
                           d_alpha, d_rgb_raw, d_background, n_rays, n_active, up, t_starts, t_ends, nsr_guard.state,
                           nsr_guard.parity);
    NSR_CHECK_LAUNCH("nsr_neus_composite_backward");
    return NSR_OK;
}

extern "C" int nsr_neus_composite_backward(const int32_t *packed_info, const float *alpha, const void *rgb_raw,
                                           int rgb_is_f32, const float *weights, const float *trans,
                                           const float *background, uint32_t background_stride,
                                           const float *opacity_bg, const float *comp_rgb_full, const float *opacity,
                                           const float *gt_rgb, const float *fg_mask, const float *acc,
                                           const float *loss_weights8, float loss_scale, float *d_alpha,
                                           float *d_rgb_raw, float *d_background, uint32_t n_rays,
                                           const int32_t *n_active, void *stream)
{
    return nsr_neus_composite_backward_ex(packed_info, alpha, rgb_raw, rgb_is_f32, weights, trans, background,
                                          background_stride, opacity_bg, comp_rgb_full, opacity, gt_rgb, fg_mask, acc,
                                          loss_weights8, loss_scale, d_alpha, d_rgb_raw, d_background, n_rays, n_active,
                                          nullptr, nullptr, nullptr, stream);
}

extern "C" int nsr_neus_shade_backward(const float *sdf_out, const float *grad, const float *normal, const float *dirs,
                                       const float *t_starts, const float *t_ends, const float *inv_s,
                                       float cos_anneal_ratio, const float *laplace, float eps, float radius,
                                       const float *d_alpha, const float *d_tex_in, uint32_t n_feat,
                                       const float *loss_weights8, float loss_scale, float n_samples, float *d_out,
                                       float *gx, float *p_in, uint32_t p_stride, float *d_taps, float *acc, uint32_t n,
                                       const int32_t *n_dev, void *stream)
{
    return nsr_neus_shade_backward_ex(sdf_out, grad, normal, dirs, t_starts, t_ends, inv_s, cos_anneal_ratio, laplace, eps,
                                      radius, d_alpha, d_tex_in, n_feat, loss_weights8, loss_scale, n_samples, d_out, gx, p_in,
                                      p_stride, d_taps, acc, n, n_dev, nullptr, stream);
}

extern "C" int nsr_neus_shade_backward_ex(const float *sdf_out, const float *grad, const float *normal, const float *dirs,
                                          const float *t_starts, const float *t_ends, const float *inv_s,
                                          float cos_anneal_ratio, const float *laplace, float eps, float radius,
                                          const float *d_alpha, const float *d_tex_in, uint32_t n_feat,
                                          const float *loss_weights8, float loss_scale, float n_samples, float *d_out,
                                          float *gx, float *p_in, uint32_t p_stride, float *d_taps, float *acc, uint32_t n,
                                          const int32_t *n_dev, const NsrNeusUpstream *upstream, void *stream)
{
    if (n == 0) return NSR_OK;
    const NeusUpstream up = make_upstream(upstream);
    const bool fd = d_taps != nullptr;
    NSR_REQUIRE(sdf_out && grad && normal && dirs && t_starts && t_ends && inv_s && d_alpha && d_tex_in && loss_weights8 &&
                    d_out && acc, "nsr_neus_shade_backward: NULL pointer");
    NSR_REQUIRE(fd ? (laplace && eps > 0.f) : (gx && p_in && p_stride >= 3),
                "nsr_neus_shade_backward: finite differences need laplace/eps/d_taps, analytic needs gx/p_in");
    NeusLossWeights lw;
    memcpy(&lw, loss_weights8, sizeof(lw));
    if (fd)
        hipLaunchKernelGGL(k_neus_shade_bwd<true>, EW_GRID_CAPPED(n), sdf_out, grad, normal, dirs, t_starts, t_ends, inv_s,
                           cos_anneal_ratio, laplace, eps, radius, d_alpha, d_tex_in, n_feat, lw, loss_scale, n_samples,
                           d_out, gx, p_in, p_stride, d_taps, acc, n, n_dev, up, nsr_guard.state, nsr_guard.parity);
    else
        hipLaunchKernelGGL(k_neus_shade_bwd<false>, EW_GRID_CAPPED(n), sdf_out, grad, normal, dirs, t_starts, t_ends, inv_s,
                           cos_anneal_ratio, laplace, eps, radius, d_alpha, d_tex_in, n_feat, lw, loss_scale, n_samples,
                           d_out, gx, p_in, p_stride, d_taps, acc, n, n_dev, up, nsr_guard.state, nsr_guard.parity);
    NSR_CHECK_LAUNCH("nsr_neus_shade_backward");
    return NSR_OK;
}

extern "C" int nsr_bg_visibility_prefix(const float *out16, float density_bias, const float *t_starts, const float *t_ends,
                                        const int32_t *packed_info, float early_stop_eps, int32_t *kept_counts,
                                        uint32_t n_rays, void *stream)
{
    if (n_rays == 0) return NSR_OK;
    NSR_REQUIRE(packed_info && kept_counts, "nsr_bg_visibility_prefix: NULL pointer");
    hipLaunchKernelGGL(k_bg_visibility, RAY_GRID(n_rays), out16, density_bias, t_starts, t_ends, packed_info,
                       early_stop_eps, kept_counts, n_rays);
    NSR_CHECK_LAUNCH("nsr_bg_visibility_prefix");
    return NSR_OK;
}

extern "C" int nsr_bg_texture_input(const float *out16, uint32_t n_feat, const float *rays_d, const int64_t *ray_indices,
                                    float *tex_in, uint32_t stride, uint32_t n, const int32_t *n_dev, void *stream)
{
    if (n == 0) return NSR_OK;
    NSR_REQUIRE(out16 && rays_d && ray_indices && tex_in, "nsr_bg_texture_input: NULL pointer");
    NSR_REQUIRE(n_feat >= 1 && n_feat <= 16 && n_feat + 16 <= stride, "nsr_bg_texture_input: n_feat=%u stride=%u", n_feat, stride);
    hipLaunchKernelGGL(k_bg_texture_input, EW_GRID(n), out16, n_feat, rays_d, ray_indices, tex_in, stride, n, n_dev);
    NSR_CHECK_LAUNCH("nsr_bg_texture_input");
    return NSR_OK;
}

extern "C" int nsr_bg_composite_forward(const int32_t *packed_info, const float *out16, float density_bias,
                                        const float *rgb_raw, const float *t_starts, const float *t_ends,
                                        const float *background, float *weights, float *trans, float *comp_rgb,
                                        float *opacity, float *depth, uint32_t n_rays, void *stream)
{
    if (n_rays == 0) return NSR_OK;
    NSR_REQUIRE(packed_info && background && weights && trans && comp_rgb && opacity && depth,
                "nsr_bg_composite_forward: NULL pointer");
    hipLaunchKernelGGL(k_bg_composite_fwd, RAY_GRID(n_rays), packed_info, out16, density_bias, rgb_raw, t_starts, t_ends,
                       background, weights, trans, comp_rgb, opacity, depth, n_rays);
    NSR_CHECK_LAUNCH("nsr_bg_composite_forward");
    return NSR_OK;
}

extern "C" int nsr_bg_composite_backward(const int32_t *packed_info, const float *out16, float density_bias,
                                         const float *rgb_raw, const float *weights, const float *trans,
                                         const float *t_starts, const float *t_ends, const float *background,
                                         const float *d_comp_rgb, float *d_logit, float *d_rgb_raw, uint32_t n_rays,
                                         void *stream)
{
    if (n_rays == 0) return NSR_OK;
    NSR_REQUIRE(packed_info && background && weights && trans && d_comp_rgb, "nsr_bg_composite_backward: NULL pointer");
    hipLaunchKernelGGL(k_bg_composite_bwd, RAY_GRID(n_rays), packed_info, out16, density_bias, rgb_raw, weights, trans,
                       t_starts, t_ends, background, d_comp_rgb, d_logit, d_rgb_raw, n_rays);
    NSR_CHECK_LAUNCH("nsr_bg_composite_backward");
    return NSR_OK;
}

extern "C" int nsr_bg_join_gradients(const float *d_logit, const float *d_tex_in, uint32_t stride, uint32_t n_feat,
                                     float *d_out16, uint32_t n, const int32_t *n_dev, void *stream)
{
    if (n == 0) return NSR_OK;
    NSR_REQUIRE(d_logit && d_tex_in && d_out16 && n_feat >= 1 && n_feat <= 16 && stride >= n_feat,
                "nsr_bg_join_gradients: bad arguments");
    hipLaunchKernelGGL(k_bg_join, EW_GRID(n), d_logit, d_tex_in, stride, n_feat, d_out16, n, n_dev);
    NSR_CHECK_LAUNCH("nsr_bg_join_gradients");
    return NSR_OK;
}

// inv_s = exp(10 variance) (models/neus.py:27-32) and d loss / d variance from the accumulator slot the shade backward
// fills: two one-thread kernels instead of ~8 elementwise launches
namespace {
// occupancy statistic of the NeuS grid refresh (models/neus.py:90-101): the alpha one marching step would get at a flat SDF
__global__ void __launch_bounds__(EW_BLOCK)
k_neus_occupancy(const float *__restrict__ sdf16, const float *__restrict__ inv_s_p, float step, float *__restrict__ occ,
                 uint32_t n, const int32_t *__restrict__ n_dev)
{
    const uint32_t i = blockIdx.x * EW_BLOCK + threadIdx.x;
    if (i >= live_count(n, n_dev)) return;
    const float inv_s = fminf(fmaxf(inv_s_p[0], 1e-6f), 1e6f), sdf = sdf16[16ull * i], h = step * 0.5f;
    const float prev = sigmoidf((sdf + h) * inv_s), next = sigmoidf((sdf - h) * inv_s);
    occ[i] = fminf(fmaxf(((prev - next) + 1e-5f) / (prev + 1e-5f), 0.f), 1.f);
}
__global__ void k_neus_inv_s(const float *__restrict__ variance, float *__restrict__ inv_s) { inv_s[0] = expf(variance[0] * 10.f); }
__global__ void k_neus_variance_grad(const float *__restrict__ acc, const float *__restrict__ inv_s,
                                     float *__restrict__ grad, int accumulate)
{
    const float g = acc[ACC_INV_S_GRAD] * inv_s[0] * 10.f;
    grad[0] = accumulate ? grad[0] + g : g;
}
}  // namespace

extern "C" int nsr_neus_inv_s(const float *variance, float *inv_s, void *stream)
{
    NSR_REQUIRE(variance && inv_s, "nsr_neus_inv_s: NULL pointer");
    hipLaunchKernelGGL(k_neus_inv_s, dim3(1), dim3(1), 0, (hipStream_t)stream, variance, inv_s);
    NSR_CHECK_LAUNCH("nsr_neus_inv_s");
    return NSR_OK;
}

extern "C" int nsr_neus_variance_gradient(const float *acc, const float *inv_s, float *grad_variance, int accumulate,
                                          void *stream)
{
    NSR_REQUIRE(acc && inv_s && grad_variance, "nsr_neus_variance_gradient: NULL pointer");
    hipLaunchKernelGGL(k_neus_variance_grad, dim3(1), dim3(1), 0, (hipStream_t)stream, acc, inv_s, grad_variance,
                       accumulate);
    NSR_CHECK_LAUNCH("nsr_neus_variance_gradient");
    return NSR_OK;
}

extern "C" int nsr_neus_occupancy_values(const float *sdf_out, const float *inv_s, float step_size, float *occ, uint32_t n,
                                         const int32_t *n_dev, void *stream)
{
    if (n == 0) return NSR_OK;
    NSR_REQUIRE(sdf_out && inv_s && occ, "nsr_neus_occupancy_values: NULL pointer");
    hipLaunchKernelGGL(k_neus_occupancy, EW_GRID(n), sdf_out, inv_s, step_size, occ, n, n_dev);
    NSR_CHECK_LAUNCH("nsr_neus_occupancy_values");
    return NSR_OK;
}
