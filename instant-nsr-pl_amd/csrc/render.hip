// Segmented transmittance scans and per-ray accumulation for gfx950 (replaces nerfacc 0.3.3
// render_weight_from_density / render_weight_from_alpha / render_visibility / accumulate_along_rays;
// reference call sites models/nerf.py:105-108, models/neus.py:181-184,237-242).
//
// nerfacc runs device-wide CUB scans keyed by ray index.  Samples of a ray are contiguous and rays are
// short (<= ~1024 samples), so the MI355X design is ONE WAVEFRONT PER RAY: the 64 lanes stride through the
// ray's segment in 64-sample chunks (coalesced 256-B loads), do the prefix scan with cross-lane shuffles
// (DPP row/bank shifts; no LDS, no device-wide pass, no temp storage) and carry the running value in a
// register.  8192 rays = 8192 waves = 32 per CU: the chip is full and every segment is deterministic
// (fixed summation order, unlike atomics or a device-wide decoupled look-back).
#include "nsr_common.h"

namespace {

constexpr int R_BLOCK = 256;         // 4 waves = 4 rays per block
constexpr int RAYS_PER_BLOCK = R_BLOCK / NSR_WAVE;
constexpr int EW_BLOCK = 256;

__device__ __forceinline__ bool wave_ray(const int32_t *__restrict__ packed, uint32_t n_rays, uint32_t &start,
                                         uint32_t &count)
{
    const uint32_t r = blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
    if (r >= n_rays) return false;
    start = (uint32_t)packed[2ull * r];
    count = (uint32_t)packed[2ull * r + 1];
    return count > 0;
}

// T_i = exp(-sum_{j<i} sigma_j dt_j)
__global__ void __launch_bounds__(R_BLOCK)
k_trans_sigma_fwd(const int32_t *__restrict__ packed, const float *__restrict__ t0, const float *__restrict__ t1,
                  const float *__restrict__ sigma, float *__restrict__ trans, uint32_t n_rays)
{
    uint32_t start, count;
    if (!wave_ray(packed, n_rays, start, count)) return;
    const uint32_t lane = threadIdx.x & 63;
    float carry = 0.f;
    for (uint32_t c = 0; c < count; c += 64) {
        const uint32_t k = c + lane;
        const bool ok = k < count;
        const float v = ok ? sigma[start + k] * (t1[start + k] - t0[start + k]) : 0.f;
        const float inc = wave_incl_scan_add(v);
        // exclusive prefix by shuffle, not as `inc - v`: an overflowed density (sigma = inf) would give inf - inf = NaN here,
        // where nerfacc's sequential loop gives T = 0 behind it
        float exc = __shfl_up(inc, 1, 64);
        if (lane == 0) exc = 0.f;
        if (ok) trans[start + k] = expf(-(carry + exc));
        carry += __shfl(inc, 63, 64);
    }
}

// grad_(sigma)_j = -dt_j * sum_{i>j} gT_i T_i   (reverse exclusive segmented sum)
__global__ void __launch_bounds__(R_BLOCK)
k_trans_sigma_bwd(const int32_t *__restrict__ packed, const float *__restrict__ t0, const float *__restrict__ t1,
                  const float *__restrict__ trans, const float *__restrict__ g_trans, float *__restrict__ g_sigma,
                  uint32_t n_rays)
{
    uint32_t start, count;
    if (!wave_ray(packed, n_rays, start, count)) return;
    const uint32_t lane = threadIdx.x & 63;
    float carry = 0.f;
    for (uint32_t c = 0; c < count; c += 64) {  // walk the segment from its END
        const uint32_t k = c + lane;
        const bool ok = k < count;
        const uint32_t idx = start + count - 1 - k;
        const float v = ok ? g_trans[idx] * trans[idx] : 0.f;
        const float inc = wave_incl_scan_add(v);
        if (ok) g_sigma[idx] = -(carry + (inc - v)) * (t1[idx] - t0[idx]);
        carry += __shfl(inc, 63, 64);
    }
}

// T_i = prod_{j<i} (1 - alpha_j)
__global__ void __launch_bounds__(R_BLOCK)
k_trans_alpha_fwd(const int32_t *__restrict__ packed, const float *__restrict__ alpha, float *__restrict__ trans,
                  uint32_t n_rays)
{
    uint32_t start, count;
    if (!wave_ray(packed, n_rays, start, count)) return;
    const uint32_t lane = threadIdx.x & 63;
    float carry = 1.f;
    for (uint32_t c = 0; c < count; c += 64) {
        const uint32_t k = c + lane;
        const bool ok = k < count;
        const float v = ok ? 1.f - alpha[start + k] : 1.f;
        const float inc = wave_incl_scan_mul(v);
        // exclusive = product of the lanes strictly before this one
        float exc = __shfl_up(inc, 1, 64);
        if (lane == 0) exc = 1.f;
        if (ok) trans[start + k] = carry * exc;
        carry *= __shfl(inc, 63, 64);
    }
}

__global__ void __launch_bounds__(R_BLOCK)
k_trans_alpha_bwd(const int32_t *__restrict__ packed, const float *__restrict__ alpha, const float *__restrict__ trans,
                  const float *__restrict__ g_trans, float *__restrict__ g_alpha, uint32_t n_rays)
{
    uint32_t start, count;
    if (!wave_ray(packed, n_rays, start, count)) return;
    const uint32_t lane = threadIdx.x & 63;
    float carry = 0.f;
    for (uint32_t c = 0; c < count; c += 64) {
        const uint32_t k = c + lane;
        const bool ok = k < count;
        const uint32_t idx = start + count - 1 - k;
        const float v = ok ? g_trans[idx] * trans[idx] : 0.f;
        const float inc = wave_incl_scan_add(v);
        if (ok) g_alpha[idx] = -(carry + (inc - v)) / fmaxf(1.f - alpha[idx], 1e-10f);
        carry += __shfl(inc, 63, 64);
    }
}

// out[r, :] = sum_i w_i * v_i[:]   (dim <= 4 handled in registers; generic dims loop)
__global__ void __launch_bounds__(R_BLOCK)
k_accumulate_fwd(const int32_t *__restrict__ packed, const float *__restrict__ w, const float *__restrict__ values,
                 uint32_t dim, float *__restrict__ out, uint32_t n_rays)
{
    const uint32_t r = blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6);
    if (r >= n_rays) return;
    const uint32_t start = (uint32_t)packed[2ull * r], count = (uint32_t)packed[2ull * r + 1];
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t d0 = 0; d0 < dim; d0 += 4) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const uint32_t nd = min(4u, dim - d0);
        for (uint32_t k = lane; k < count; k += 64) {
            const float wk = w[start + k];
            for (uint32_t d = 0; d < nd; ++d)
                acc[d] += values ? wk * values[(uint64_t)(start + k) * dim + d0 + d] : wk;
        }
        for (uint32_t d = 0; d < nd; ++d) {
            const float s = wave_sum(acc[d]);
            if (lane == 0) out[(uint64_t)r * dim + d0 + d] = s;
        }
    }
}

__global__ void __launch_bounds__(EW_BLOCK)
k_accumulate_bwd(const int64_t *__restrict__ ray_indices, const float *__restrict__ w, const float *__restrict__ values,
                 uint32_t dim, const float *__restrict__ g_out, float *__restrict__ g_w, float *__restrict__ g_v,
                 uint32_t n)
{
    const uint32_t i = blockIdx.x * EW_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int64_t r = ray_indices[i];
    const float wi = w[i];
    float gw = 0.f;
    for (uint32_t d = 0; d < dim; ++d) {
        const float go = g_out[(uint64_t)r * dim + d];
        if (values) {
            gw = fmaf(go, values[(uint64_t)i * dim + d], gw);
            if (g_v) g_v[(uint64_t)i * dim + d] = wi * go;
        } else {
            gw += go;
        }
    }
    if (g_w) g_w[i] = gw;
}

// ---- distortion loss (Mip-NeRF 360; the reference takes torch_efficient_distloss.flatten_eff_distloss,
// systems/nerf.py:103-106, systems/neus.py:131-139) -------------------------------------------------------------------
// Per ray, for samples sorted along the ray:  L = sum_i sum_j w_i w_j |m_i - m_j| + 1/3 sum_i w_i^2 dt_i
//                                               = sum_i [ 2 w_i (m_i W_<i - WM_<i) + 1/3 dt_i w_i^2 ]
// with the exclusive prefix sums W_<i = sum_{j<i} w_j and WM_<i = sum_{j<i} w_j m_j.  One wavefront per ray, wave scans
// with a running carry, no atomics.  ray_loss[r] <- L of ray r (0 for empty rays).
__global__ void __launch_bounds__(R_BLOCK)
k_distortion_fwd(const int32_t *__restrict__ packed, const float *__restrict__ w, const float *__restrict__ m,
                 const float *__restrict__ dt, float *__restrict__ ray_loss, uint32_t n_rays)
{
    const uint32_t r = blockIdx.x * RAYS_PER_BLOCK + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= n_rays) return;
    const uint32_t start = (uint32_t)packed[2ull * r], count = (uint32_t)packed[2ull * r + 1];
    float cw = 0.f, cwm = 0.f, acc = 0.f;
    for (uint32_t c = 0; c < count; c += 64) {
        const uint32_t k = c + lane;
        const bool ok = k < count;
        const float wi = ok ? w[start + k] : 0.f, mi = ok ? m[start + k] : 0.f, di = ok ? dt[start + k] : 0.f;
        const float iw = wave_incl_scan_add(wi), iwm = wave_incl_scan_add(wi * mi);
        const float w_pre = cw + (iw - wi), wm_pre = cwm + (iwm - wi * mi);
        acc += 2.f * wi * (mi * w_pre - wm_pre) + (1.f / 3.f) * di * wi * wi;
        cw += __shfl(iw, 63, 64);
        cwm += __shfl(iwm, 63, 64);
    }
    acc = wave_sum(acc);
    if (lane == 0) ray_loss[r] = acc;
}

// dL/dw_i = 2 [ m_i (W_<i - W_>i) + (WM_>i - WM_<i) ] + 2/3 dt_i w_i      (suffix = total - prefix - own)
__global__ void __launch_bounds__(R_BLOCK)
k_distortion_bwd(const int32_t *__restrict__ packed, const float *__restrict__ w, const float *__restrict__ m,
                 const float *__restrict__ dt, float *__restrict__ grad_w, uint32_t n_rays)
{
    uint32_t start, count;
    if (!wave_ray(packed, n_rays, start, count)) return;
    const uint32_t lane = threadIdx.x & 63;
    float tw = 0.f, twm = 0.f;
    for (uint32_t k = lane; k < count; k += 64) { tw += w[start + k]; twm += w[start + k] * m[start + k]; }
    tw = wave_sum(tw);
    twm = wave_sum(twm);
    float cw = 0.f, cwm = 0.f;
    for (uint32_t c = 0; c < count; c += 64) {
        const uint32_t k = c + lane;
        const bool ok = k < count;
        const float wi = ok ? w[start + k] : 0.f, mi = ok ? m[start + k] : 0.f, di = ok ? dt[start + k] : 0.f;
        const float iw = wave_incl_scan_add(wi), iwm = wave_incl_scan_add(wi * mi);
        const float w_pre = cw + (iw - wi), wm_pre = cwm + (iwm - wi * mi);
        const float w_suf = tw - (w_pre + wi), wm_suf = twm - (wm_pre + wi * mi);
        if (ok) grad_w[start + k] = 2.f * (mi * (w_pre - w_suf) + (wm_suf - wm_pre)) + (2.f / 3.f) * di * wi;
        cw += __shfl(iw, 63, 64);
        cwm += __shfl(iwm, 63, 64);
    }
}

}  // namespace

#define RAY_GRID(n_rays) dim3(nsr_div_up(n_rays, RAYS_PER_BLOCK)), dim3(R_BLOCK), 0, (hipStream_t)stream

extern "C" int nsr_transmittance_from_sigma_forward(const int32_t *packed_info, const float *t_starts,
                                                    const float *t_ends, const float *sigmas, float *trans,
                                                    uint32_t n_rays, void *stream)
{
    if (n_rays == 0) return NSR_OK;
    NSR_REQUIRE(packed_info, "nsr_transmittance_from_sigma_forward: NULL packed_info");
    hipLaunchKernelGGL(k_trans_sigma_fwd, RAY_GRID(n_rays), packed_info, t_starts, t_ends, sigmas, trans, n_rays);
    NSR_CHECK_LAUNCH("nsr_transmittance_from_sigma_forward");
    return NSR_OK;
}

extern "C" int nsr_transmittance_from_sigma_backward(const int32_t *packed_info, const float *t_starts,
                                                     const float *t_ends, const float *trans, const float *grad_trans,
                                                     float *grad_sigmas, uint32_t n_rays, void *stream)
{
    if (n_rays == 0) return NSR_OK;
    NSR_REQUIRE(packed_info, "nsr_transmittance_from_sigma_backward: NULL packed_info");
    hipLaunchKernelGGL(k_trans_sigma_bwd, RAY_GRID(n_rays), packed_info, t_starts, t_ends, trans, grad_trans,
                       grad_sigmas, n_rays);
    NSR_CHECK_LAUNCH("nsr_transmittance_from_sigma_backward");
    return NSR_OK;
}

extern "C" int nsr_transmittance_from_alpha_forward(const int32_t *packed_info, const float *alphas, float *trans,
                                                    uint32_t n_rays, void *stream)
{
    if (n_rays == 0) return NSR_OK;
    NSR_REQUIRE(packed_info, "nsr_transmittance_from_alpha_forward: NULL packed_info");
    hipLaunchKernelGGL(k_trans_alpha_fwd, RAY_GRID(n_rays), packed_info, alphas, trans, n_rays);
    NSR_CHECK_LAUNCH("nsr_transmittance_from_alpha_forward");
    return NSR_OK;
}

extern "C" int nsr_transmittance_from_alpha_backward(const int32_t *packed_info, const float *alphas,
                                                     const float *trans, const float *grad_trans, float *grad_alphas,
                                                     uint32_t n_rays, void *stream)
{
    if (n_rays == 0) return NSR_OK;
    NSR_REQUIRE(packed_info, "nsr_transmittance_from_alpha_backward: NULL packed_info");
    hipLaunchKernelGGL(k_trans_alpha_bwd, RAY_GRID(n_rays), packed_info, alphas, trans, grad_trans, grad_alphas,
                       n_rays);
    NSR_CHECK_LAUNCH("nsr_transmittance_from_alpha_backward");
    return NSR_OK;
}

extern "C" int nsr_accumulate_along_rays_forward(const int32_t *packed_info, const float *weights,
                                                 const float *values, uint32_t dim, float *out, uint32_t n_rays,
                                                 void *stream)
{
    if (n_rays == 0) return NSR_OK;
    NSR_REQUIRE(packed_info && out && dim >= 1, "nsr_accumulate_along_rays_forward: bad arguments");
    hipLaunchKernelGGL(k_accumulate_fwd, RAY_GRID(n_rays), packed_info, weights, values, dim, out, n_rays);
    NSR_CHECK_LAUNCH("nsr_accumulate_along_rays_forward");
    return NSR_OK;
}

extern "C" int nsr_accumulate_along_rays_backward(const int64_t *ray_indices, const float *weights,
                                                  const float *values, uint32_t dim, const float *grad_out,
                                                  float *grad_weights, float *grad_values, uint32_t n, void *stream)
{
    if (n == 0) return NSR_OK;
    NSR_REQUIRE(ray_indices && weights && grad_out && dim >= 1, "nsr_accumulate_along_rays_backward: bad arguments");
    hipLaunchKernelGGL(k_accumulate_bwd, dim3(nsr_div_up(n, EW_BLOCK)), dim3(EW_BLOCK), 0, (hipStream_t)stream,
                       ray_indices, weights, values, dim, grad_out, grad_weights, grad_values, n);
    NSR_CHECK_LAUNCH("nsr_accumulate_along_rays_backward");
    return NSR_OK;
}

extern "C" int nsr_distortion_loss_forward(const int32_t *packed_info, const float *weights, const float *midpoints,
                                           const float *intervals, float *ray_loss, uint32_t n_rays, void *stream)
{
    if (n_rays == 0) return NSR_OK;
    NSR_REQUIRE(packed_info && ray_loss, "nsr_distortion_loss_forward: NULL pointer");
    hipLaunchKernelGGL(k_distortion_fwd, RAY_GRID(n_rays), packed_info, weights, midpoints, intervals, ray_loss, n_rays);
    NSR_CHECK_LAUNCH("nsr_distortion_loss_forward");
    return NSR_OK;
}

extern "C" int nsr_distortion_loss_backward(const int32_t *packed_info, const float *weights, const float *midpoints,
                                            const float *intervals, float *grad_weights, uint32_t n_rays, void *stream)
{
    if (n_rays == 0) return NSR_OK;
    NSR_REQUIRE(packed_info && grad_weights, "nsr_distortion_loss_backward: NULL pointer");
    hipLaunchKernelGGL(k_distortion_bwd, RAY_GRID(n_rays), packed_info, weights, midpoints, intervals, grad_weights,
                       n_rays);
    NSR_CHECK_LAUNCH("nsr_distortion_loss_backward");
    return NSR_OK;
}
