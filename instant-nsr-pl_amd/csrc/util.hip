// Error reporting + the small fused elementwise pieces of the reference glue.
#include <stdarg.h>
#include <stdio.h>

#include "nsr_common.h"

static thread_local char g_err[512] = "";

thread_local hipEvent_t nsr_next_stop_event = nullptr;

void nsr_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *nsr_last_error(void) { return g_err; }
extern "C" int nsr_abi_version(void) { return 1; }

namespace {

constexpr int EW_BLOCK = 256;

// ---- spherical harmonics, degree 4 (tcnn SphericalHarmonics; reference models/texture.py:25) ----
__global__ void __launch_bounds__(EW_BLOCK)
k_sh4(const float *__restrict__ u, __half *__restrict__ out, uint32_t n, uint32_t stride)
{
    const uint32_t i = blockIdx.x * EW_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float x = u[3ull * i] * 2.f - 1.f, y = u[3ull * i + 1] * 2.f - 1.f, z = u[3ull * i + 2] * 2.f - 1.f;
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
    __half *p = out + (uint64_t)i * stride;
    if ((stride & 7u) == 0) {  // 16-byte aligned rows: two 16-B stores
        __half2 h[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) h[k] = __floats2half2_rn(o[2 * k], o[2 * k + 1]);
        reinterpret_cast<uint4 *>(p)[0] = *reinterpret_cast<uint4 *>(&h[0]);
        reinterpret_cast<uint4 *>(p)[1] = *reinterpret_cast<uint4 *>(&h[4]);
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) p[k] = __float2half_rn(o[k]);
    }
}

// ---- contract_to_unisphere (reference models/geometry.py:17-29) ----------------------------------
__global__ void __launch_bounds__(EW_BLOCK)
k_contract_to_unisphere(const float *__restrict__ x, float radius, int type, float *__restrict__ out, uint32_t n)
{
    const uint32_t i = blockIdx.x * EW_BLOCK + threadIdx.x;
    if (i >= n) return;
    // scale_anything(x, (-r, r), (0, 1)) = (x + r) / (2r) * 1 + 0
    const float den = radius - (-radius);
    float v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = (x[3ull * i + k] - (-radius)) / den;
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
    for (int k = 0; k < 3; ++k) out[3ull * i + k] = v[k];
}

// ---- density = exp(out[:,0] + bias); feature = out (fp32) (reference models/geometry.py:124-129) --
__global__ void __launch_bounds__(EW_BLOCK)
k_density_activation(const __half *__restrict__ mlp_out, uint32_t stride, uint32_t n_feat, float bias,
                     float *__restrict__ density, float *__restrict__ feature, uint32_t n)
{
    const uint32_t i = blockIdx.x * EW_BLOCK + threadIdx.x;
    if (i >= n) return;
    const __half *p = mlp_out + (uint64_t)i * stride;
    density[i] = expf(__half2float(p[0]) + bias);
    if (feature)
        for (uint32_t k = 0; k < n_feat; ++k) feature[(uint64_t)i * n_feat + k] = __half2float(p[k]);
}

// ---- sample positions: p = o[r] + d[r] * (t0 + t1) / 2 (reference models/nerf.py:95-99) ----------
__global__ void __launch_bounds__(EW_BLOCK)
k_sample_positions(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                   const int64_t *__restrict__ ray_indices, const float *__restrict__ t0,
                   const float *__restrict__ t1, float *__restrict__ pos, float *__restrict__ dirs, uint32_t n)
{
    const uint32_t i = blockIdx.x * EW_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int64_t r = ray_indices[i];
    const float tm = (t0[i] + t1[i]) / 2.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float d = rays_d[3 * r + k];
        // torch evaluates t_dirs * midpoints then adds the origin: two roundings, no fma
        pos[3ull * i + k] = __fadd_rn(rays_o[3 * r + k], __fmul_rn(d, tm));
        if (dirs) dirs[3ull * i + k] = d;
    }
}

// ---- NeuS SDF -> alpha (reference models/neus.py:117-139) ----------------------------------------
__device__ __forceinline__ float sigmoidf(float v) { return 1.f / (1.f + expf(-v)); }

__global__ void __launch_bounds__(EW_BLOCK)
k_neus_alpha_fwd(const float *__restrict__ sdf, const float *__restrict__ normal, const float *__restrict__ dirs,
                 const float *__restrict__ dists, const float *__restrict__ inv_s_p, float anneal,
                 float *__restrict__ alpha, uint32_t n)
{
    const uint32_t i = blockIdx.x * EW_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float inv_s = fminf(fmaxf(inv_s_p[0], 1e-6f), 1e6f);
    const float tc = dirs[3ull * i] * normal[3ull * i] + dirs[3ull * i + 1] * normal[3ull * i + 1] +
                     dirs[3ull * i + 2] * normal[3ull * i + 2];
    const float ic = -(fmaxf(-tc * 0.5f + 0.5f, 0.f) * (1.f - anneal) + fmaxf(-tc, 0.f) * anneal);
    const float h = ic * dists[i] * 0.5f;
    const float prev = sigmoidf((sdf[i] - h) * inv_s), next = sigmoidf((sdf[i] + h) * inv_s);
    const float a = ((prev - next) + 1e-5f) / (prev + 1e-5f);
    alpha[i] = fminf(fmaxf(a, 0.f), 1.f);
}

__global__ void __launch_bounds__(EW_BLOCK)
k_neus_alpha_bwd(const float *__restrict__ sdf, const float *__restrict__ normal, const float *__restrict__ dirs,
                 const float *__restrict__ dists, const float *__restrict__ inv_s_p, float anneal,
                 const float *__restrict__ g_alpha, float *__restrict__ g_sdf, float *__restrict__ g_normal,
                 float *__restrict__ g_inv_s, uint32_t n)
{
    const uint32_t i = blockIdx.x * EW_BLOCK + threadIdx.x;
    float gs_local = 0.f;
    if (i < n) {
        const float raw_inv_s = inv_s_p[0];
        const float inv_s = fminf(fmaxf(raw_inv_s, 1e-6f), 1e6f);
        const bool s_live = raw_inv_s >= 1e-6f && raw_inv_s <= 1e6f;
        const float d0 = dirs[3ull * i], d1 = dirs[3ull * i + 1], d2 = dirs[3ull * i + 2];
        const float tc = d0 * normal[3ull * i] + d1 * normal[3ull * i + 1] + d2 * normal[3ull * i + 2];
        const float u1 = -tc * 0.5f + 0.5f, u2 = -tc;
        const float ic = -(fmaxf(u1, 0.f) * (1.f - anneal) + fmaxf(u2, 0.f) * anneal);
        // d ic / d tc
        const float dic_dtc = -((u1 > 0.f ? -0.5f : 0.f) * (1.f - anneal) + (u2 > 0.f ? -1.f : 0.f) * anneal);
        const float dist = dists[i], s = sdf[i];
        const float h = ic * dist * 0.5f;
        const float ep = (s - h), en = (s + h);
        const float prev = sigmoidf(ep * inv_s), next = sigmoidf(en * inv_s);
        const float num = (prev - next) + 1e-5f, den = prev + 1e-5f;
        const float a = num / den;
        float ga = g_alpha[i];
        if (!(a >= 0.f && a <= 1.f)) ga = 0.f;  // clip(0,1) kills the gradient outside
        // a = num/den ; d a/d prev = (den - num)/den^2 ; d a/d next = -1/den
        const float da_dprev = (den - num) / (den * den), da_dnext = -1.f / den;
        const float gp = ga * da_dprev * prev * (1.f - prev);  // grad w.r.t. (ep*inv_s)
        const float gn = ga * da_dnext * next * (1.f - next);  // grad w.r.t. (en*inv_s)
        if (g_sdf) g_sdf[i] = (gp + gn) * inv_s;
        const float g_h = (-gp + gn) * inv_s;  // ep = s-h, en = s+h
        const float g_tc = g_h * dist * 0.5f * dic_dtc;
        if (g_normal) {
            g_normal[3ull * i] = g_tc * d0; g_normal[3ull * i + 1] = g_tc * d1; g_normal[3ull * i + 2] = g_tc * d2;
        }
        gs_local = s_live ? (gp * ep + gn * en) : 0.f;
    }
    if (g_inv_s) {
        gs_local = wave_sum(gs_local);
        if ((threadIdx.x & 63) == 0 && gs_local != 0.f) unsafeAtomicAdd(g_inv_s, gs_local);
    }
}

// ---- fused AdamW (torch.optim.AdamW semantics) + fp16 shadow refresh + grad zeroing --------------
__global__ void __launch_bounds__(EW_BLOCK)
k_adamw(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
        __half *__restrict__ shadow, uint64_t n, float lr, float b1, float b2, float eps, float wd, float bc1,
        float bc2, float unscale, int zero_grad, const float *__restrict__ hyper, uint64_t zero_first_n,
        const int32_t *__restrict__ skip /* overflow guard: this step's found-inf flag, or NULL */)
{
    if (skip && *skip) return;
    if (hyper) { lr = hyper[0]; bc1 = hyper[1]; bc2 = hyper[2]; }  // device-side schedule (nsr_adam_tick): graph-replayable
    const uint64_t stride = (uint64_t)gridDim.x * EW_BLOCK * 4;
    for (uint64_t base = ((uint64_t)blockIdx.x * EW_BLOCK + threadIdx.x) * 4; base < n; base += stride) {
        if (base + 4 <= n) {
            float4 pp = *reinterpret_cast<float4 *>(p + base), gg = *reinterpret_cast<float4 *>(g + base);
            float4 mm = *reinterpret_cast<float4 *>(m + base), vv = *reinterpret_cast<float4 *>(v + base);
            float *pa = &pp.x, *ga = &gg.x, *ma = &mm.x, *va = &vv.x;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                nsr_adamw_elem(pa[k], ma[k], va[k], ga[k] * unscale, lr, b1, b2, eps, wd, bc1, bc2);
            }
            *reinterpret_cast<float4 *>(p + base) = pp;
            *reinterpret_cast<float4 *>(m + base) = mm;
            *reinterpret_cast<float4 *>(v + base) = vv;
            if (zero_grad && base < zero_first_n) *reinterpret_cast<float4 *>(g + base) = make_float4(0.f, 0.f, 0.f, 0.f);
            if (shadow) {
                __half2 h[2] = {__floats2half2_rn(pa[0], pa[1]), __floats2half2_rn(pa[2], pa[3])};
                *reinterpret_cast<uint2 *>(shadow + base) = *reinterpret_cast<uint2 *>(h);
            }
        } else {
            for (uint64_t j = base; j < n; ++j) {
                float pj = p[j], mj = m[j], vj = v[j];
                nsr_adamw_elem(pj, mj, vj, g[j] * unscale, lr, b1, b2, eps, wd, bc1, bc2);
                p[j] = pj; m[j] = mj; v[j] = vj;
                if (zero_grad && j < zero_first_n) g[j] = 0.f;
                if (shadow) shadow[j] = __float2half_rn(pj);
            }
        }
    }
}

// gradient transport in half precision: dst = half(src * scale) and back (dst = float(src) * scale), 8 elements per lane
__global__ void __launch_bounds__(EW_BLOCK)
k_scale_to_half(const float *__restrict__ src, __half *__restrict__ dst, uint64_t n, float scale)
{
    const uint64_t stride = (uint64_t)gridDim.x * EW_BLOCK * 8;
    for (uint64_t b = ((uint64_t)blockIdx.x * EW_BLOCK + threadIdx.x) * 8; b < n; b += stride) {
        if (b + 8 <= n) {
            const float4 a = *reinterpret_cast<const float4 *>(src + b), c = *reinterpret_cast<const float4 *>(src + b + 4);
            __half2 h[4] = {__floats2half2_rn(a.x * scale, a.y * scale), __floats2half2_rn(a.z * scale, a.w * scale),
                            __floats2half2_rn(c.x * scale, c.y * scale), __floats2half2_rn(c.z * scale, c.w * scale)};
            *reinterpret_cast<uint4 *>(dst + b) = *reinterpret_cast<uint4 *>(h);
        } else {
            for (uint64_t j = b; j < n; ++j) dst[j] = __float2half_rn(src[j] * scale);
        }
    }
}

__global__ void __launch_bounds__(EW_BLOCK)
k_scale_from_half(const __half *__restrict__ src, float *__restrict__ dst, uint64_t n, float scale)
{
    const uint64_t stride = (uint64_t)gridDim.x * EW_BLOCK * 8;
    for (uint64_t b = ((uint64_t)blockIdx.x * EW_BLOCK + threadIdx.x) * 8; b < n; b += stride) {
        if (b + 8 <= n) {
            const uint4 raw = *reinterpret_cast<const uint4 *>(src + b);
            const __half2 *h = reinterpret_cast<const __half2 *>(&raw);
            float4 a, c;
            a.x = __low2float(h[0]) * scale; a.y = __high2float(h[0]) * scale;
            a.z = __low2float(h[1]) * scale; a.w = __high2float(h[1]) * scale;
            c.x = __low2float(h[2]) * scale; c.y = __high2float(h[2]) * scale;
            c.z = __low2float(h[3]) * scale; c.w = __high2float(h[3]) * scale;
            *reinterpret_cast<float4 *>(dst + b) = a;
            *reinterpret_cast<float4 *>(dst + b + 4) = c;
        } else {
            for (uint64_t j = b; j < n; ++j) dst[j] = __half2float(src[j]) * scale;
        }
    }
}

// step counter, MultiStepLR-scaled learning rate and bias corrections on the device, in the double arithmetic the host
// path uses: a captured step (hipGraph) then needs no per-step host scalar
__global__ void k_adam_tick(int32_t *__restrict__ step, float *__restrict__ hyper, double base_lr, double b1, double b2,
                            double gamma, int32_t m0, int32_t m1, int32_t m2, const int32_t *__restrict__ guard, int parity)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    if (guard && guard[parity]) return;  // overflow guard: the step is skipped, the optimizer's step count does not advance
    const int32_t done = *step;  // optimizer steps taken so far == the trainer's global_step
    const int32_t s = done + 1;
    *step = s;
    const int k = (done >= m0) + (done >= m1) + (done >= m2);
    double scale = 1.0;
    for (int i = 0; i < k; ++i) scale *= gamma;
    // beta^s as running products kept (as doubles) in hyper[4..7]: a software pow() per step costs this one-thread
    // kernel ~10 us; restarted with pow() whenever the counter does not continue the stored one (first call, resume)
    double *pw = reinterpret_cast<double *>(hyper + 4);
    int32_t *pw_step = reinterpret_cast<int32_t *>(hyper + 3);
    double p1, p2;
    if (*pw_step == done && done > 0) { p1 = pw[0] * b1; p2 = pw[1] * b2; }
    else { p1 = pow(b1, (double)s); p2 = pow(b2, (double)s); }
    pw[0] = p1; pw[1] = p2; *pw_step = s;
    hyper[0] = (float)(base_lr * scale);
    hyper[1] = (float)(1.0 - p1);
    hyper[2] = (float)(1.0 - p2);
}

// ---- the optimizer step of the asynchronous trainer in ONE launch: nsr_adam_tick + nsr_adamw_step over up to two
// parameter tensors (hash table + density MLP, colour MLP).  Every workgroup derives (lr, bias corrections) from the
// device-side step counter itself -- same double arithmetic as k_adam_tick, read-only -- and the LAST workgroup to finish
// (ticket counter in hyper[8]) publishes the advanced counter / running beta powers.  Bit-identical to the three-launch
// sequence; removes two dependent launches from the step's critical path.
struct AdamSeg {
    float *p, *g, *m, *v;
    __half *shadow;
    uint64_t n, zero_first_n;
};

__device__ __forceinline__ void adamw_span(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m,
                                           float *__restrict__ v, __half *__restrict__ shadow, uint64_t n,
                                           uint64_t zero_first_n, float lr, float b1, float b2, float eps, float wd,
                                           float bc1, float bc2, float unscale, int zero_grad)
{
    const uint64_t stride = (uint64_t)gridDim.x * EW_BLOCK * 4;
    for (uint64_t base = ((uint64_t)blockIdx.x * EW_BLOCK + threadIdx.x) * 4; base < n; base += stride) {
        if (base + 4 <= n) {
            float4 pp = *reinterpret_cast<float4 *>(p + base), gg = *reinterpret_cast<float4 *>(g + base);
            float4 mm = *reinterpret_cast<float4 *>(m + base), vv = *reinterpret_cast<float4 *>(v + base);
            float *pa = &pp.x, *ga = &gg.x, *ma = &mm.x, *va = &vv.x;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                nsr_adamw_elem(pa[k], ma[k], va[k], ga[k] * unscale, lr, b1, b2, eps, wd, bc1, bc2);
            }
            *reinterpret_cast<float4 *>(p + base) = pp;
            *reinterpret_cast<float4 *>(m + base) = mm;
            *reinterpret_cast<float4 *>(v + base) = vv;
            if (zero_grad && base < zero_first_n) *reinterpret_cast<float4 *>(g + base) = make_float4(0.f, 0.f, 0.f, 0.f);
            if (shadow) {
                __half2 h[2] = {__floats2half2_rn(pa[0], pa[1]), __floats2half2_rn(pa[2], pa[3])};
                *reinterpret_cast<uint2 *>(shadow + base) = *reinterpret_cast<uint2 *>(h);
            }
        } else {
            for (uint64_t j = base; j < n; ++j) {
                float pj = p[j], mj = m[j], vj = v[j];
                nsr_adamw_elem(pj, mj, vj, g[j] * unscale, lr, b1, b2, eps, wd, bc1, bc2);
                p[j] = pj; m[j] = mj; v[j] = vj;
                if (zero_grad && j < zero_first_n) g[j] = 0.f;
                if (shadow) shadow[j] = __float2half_rn(pj);
            }
        }
    }
}

// a skipped step still clears the gradients it would have consumed (the weight-gradient kernels accumulate into them)
__device__ __forceinline__ void zero_span(float *__restrict__ g, uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * EW_BLOCK * 4;
    for (uint64_t base = ((uint64_t)blockIdx.x * EW_BLOCK + threadIdx.x) * 4; base < n; base += stride) {
        if (base + 4 <= n) *reinterpret_cast<float4 *>(g + base) = make_float4(0.f, 0.f, 0.f, 0.f);
        else for (uint64_t j = base; j < n; ++j) g[j] = 0.f;
    }
}

__global__ void __launch_bounds__(EW_BLOCK)
k_adamw_scheduled(AdamSeg a, AdamSeg b, int32_t *step, float *hyper /* 12 floats */, int32_t *step_out, float *hyper_out,
                  double base_lr, double b1d, double b2d, double gamma, int32_t m0, int32_t m1, int32_t m2, float b1,
                  float b2, float eps, float wd, float unscale, int zero_grad,
                  int32_t *guard /* NsrGuard state or NULL */, int parity, float scale0)
{
    // overflow guard: this step's found-inf flag and scale.  The gradients were unscaled by scale0 (the host's constant) where
    // they were reduced; the data-gradient kernel scaled by the device-side value -- the ratio is applied here.
    const bool skip = guard && guard[parity] != 0;
    if (guard) unscale *= scale0 / __int_as_float(guard[2]);
    __shared__ float hs[3];
    __shared__ double pws[2];
    // the advanced schedule state goes to (step_out, hyper_out): the same words, or the other half of a double buffer when
    // a kernel on ANOTHER stream (the table backward's fused AdamW) reads this step's state while this launch runs
    double *pw = reinterpret_cast<double *>(hyper_out + 4);
    int32_t *pw_step = reinterpret_cast<int32_t *>(hyper_out + 3);
    const int32_t s = *step + 1;
    if (threadIdx.x == 0) {
        float lr0, c1, c2;
        double p1, p2;
        nsr_adam_schedule(step, hyper, base_lr, b1d, b2d, gamma, m0, m1, m2, lr0, c1, c2, p1, p2);
        pws[0] = p1; pws[1] = p2;
        hs[0] = lr0; hs[1] = c1; hs[2] = c2;
    }
    __syncthreads();
    const float lr = hs[0], bc1 = hs[1], bc2 = hs[2];
    if (skip) {  // weights, moments and the fp16 images stay as they are
        if (zero_grad) { zero_span(a.g, a.zero_first_n < a.n ? a.zero_first_n : a.n); if (b.n) zero_span(b.g, b.n); }
    } else {
        adamw_span(a.p, a.g, a.m, a.v, a.shadow, a.n, a.zero_first_n, lr, b1, b2, eps, wd, bc1, bc2, unscale, zero_grad);
        if (b.n) adamw_span(b.p, b.g, b.m, b.v, b.shadow, b.n, b.zero_first_n, lr, b1, b2, eps, wd, bc1, bc2, unscale, zero_grad);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // no fence: the only cross-workgroup ordering needed is "every workgroup READ the old schedule state before the
        // last one overwrites it", and those loads completed long before this point (a release fence here would write
        // back + invalidate the XCD's L2 once per workgroup -- measured 2x on the whole kernel)
        uint32_t *ticket = reinterpret_cast<uint32_t *>(hyper + 8);
        if (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
            *ticket = 0;
            if (skip) {  // the optimizer step did not happen: the schedule state is carried over unchanged (GradScaler.step)
                const double *pw_in = reinterpret_cast<const double *>(hyper + 4);
                *step_out = *step;
                pw[0] = pw_in[0]; pw[1] = pw_in[1]; *pw_step = *reinterpret_cast<const int32_t *>(hyper + 3);
                hyper_out[0] = hyper[0]; hyper_out[1] = hyper[1]; hyper_out[2] = hyper[2];
            } else {
                *step_out = s;
                pw[0] = pws[0]; pw[1] = pws[1]; *pw_step = s;
                hyper_out[0] = hs[0]; hyper_out[1] = hs[1]; hyper_out[2] = hs[2];
            }
            if (guard) {  // GradScaler.update(): backoff 0.5 on overflow, growth 2 after growth_interval clean steps
                float sc = __int_as_float(guard[2]);
                if (skip) { sc = fmaxf(sc * 0.5f, 1.f); guard[3] = 0; guard[4] += 1; }
                else if (++guard[3] >= guard[5]) { sc = fminf(sc * 2.f, 1.8446744e19f); guard[3] = 0; }
                guard[2] = __float_as_int(sc);
            }
        }
    }
}

// ---- AdamW over MANY small tensors in one launch (the fp32 heads + variance of the NeuS systems: 7..19 tensors of
// 1..4096 elements, each with its parameter group's learning rate): blockIdx.y = tensor --------------------------------
constexpr int ADAM_MULTI_MAX = 32;
struct AdamMulti {
    float *p[ADAM_MULTI_MAX], *g[ADAM_MULTI_MAX], *m[ADAM_MULTI_MAX], *v[ADAM_MULTI_MAX];
    uint32_t n[ADAM_MULTI_MAX];
    float lr[ADAM_MULTI_MAX];
};

// the device-side step count of nsr_adamw_multi (step_dev / hyper_dev given): advanced -- and the bias corrections 1 - beta^step
// left in hyper[1..2] -- unless the overflow guard has flagged this step; then GradScaler.update() for the step, whose last
// optimizer launch this is (skip: scale x 0.5, else x 2 after `growth interval` clean steps; csrc/nsr_common.h NsrGuard)
__global__ void k_adam_multi_tick(int32_t *__restrict__ step, float *__restrict__ hyper, double b1, double b2,
                                  int32_t *__restrict__ guard, int parity)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const bool skip = guard && guard[parity];
    if (!skip) {
        const int32_t s = *step + 1;
        *step = s;
        hyper[1] = (float)(1.0 - pow(b1, (double)s));
        hyper[2] = (float)(1.0 - pow(b2, (double)s));
    }
    if (guard) {
        float sc = __int_as_float(guard[2]);
        if (skip) { sc = fmaxf(sc * 0.5f, 1.f); guard[3] = 0; guard[4] += 1; }
        else if (++guard[3] >= guard[5]) { sc = fminf(sc * 2.f, 1.8446744e19f); guard[3] = 0; }
        guard[2] = __float_as_int(sc);
    }
}

__global__ void __launch_bounds__(EW_BLOCK)
k_adamw_multi(const AdamMulti t, float b1, float b2, float eps, float wd, float bc1, float bc2, int zero_grad,
              const float *__restrict__ hyper /* device-side bias corrections, or NULL */,
              const int32_t *__restrict__ skip /* overflow guard: this step's found-inf flag, or NULL */)
{
    if (skip && *skip) return;
    if (hyper) { bc1 = hyper[1]; bc2 = hyper[2]; }
    const uint32_t s = blockIdx.y, n = t.n[s];
    float *p = t.p[s], *g = t.g[s], *m = t.m[s], *v = t.v[s];
    const float lr = t.lr[s];
    for (uint32_t j = blockIdx.x * EW_BLOCK + threadIdx.x; j < n; j += gridDim.x * EW_BLOCK) {
        float pj = p[j], mj = m[j], vj = v[j];
        nsr_adamw_elem(pj, mj, vj, g[j], lr, b1, b2, eps, wd, bc1, bc2);
        p[j] = pj; m[j] = mj; v[j] = vj;
        if (zero_grad) g[j] = 0.f;
    }
}


// ---- masked mean losses of the reference's systems (systems/nerf.py:97 smooth-L1, systems/neus.py:98,102 MSE / L1 over
// `x[rays_valid[..., 0]]` pairs): the boolean-mask gathers (a nonzero + a host synchronisation each in torch) never happen --
// per-block partial sums + one wave that adds them in index order (bit-reproducible), one elementwise kernel for d loss / d pred.  kind: 0 smooth-L1(beta), 1 MSE, 2 L1, 3 Huber(delta = beta) -- torch.nn.functional semantics.
constexpr uint32_t ML_MAX_BLOCKS = 256;

__device__ __forceinline__ float masked_loss_value(float d, int kind, float beta)
{
    const float a = fabsf(d);
    switch (kind) {
    case 0: return (beta > 0.f && a < beta) ? 0.5f * d * d / beta : a - 0.5f * beta;
    case 1: return d * d;
    case 2: return a;
    default: return a <= beta ? 0.5f * d * d : beta * (a - 0.5f * beta);
    }
}

__device__ __forceinline__ float masked_loss_slope(float d, int kind, float beta)
{
    const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    const float a = fabsf(d);
    switch (kind) {
    case 0: return (beta > 0.f && a < beta) ? d / beta : sg;
    case 1: return 2.f * d;
    case 2: return sg;
    default: return a <= beta ? d : beta * sg;
    }
}

// partial sums: block b owns the elements [b, b + 1) * ML_CHUNK ..., 8 per thread and pass, loads issued unconditionally (a
// branch on the mask would put two dependent memory round trips into every pass); the partials are summed by ONE wave in index
// order (k_masked_loss_finish), so the result does not depend on which block finished first.
__global__ void __launch_bounds__(EW_BLOCK)
k_masked_loss_partials(const float *__restrict__ pred, const float *__restrict__ target, const uint8_t *__restrict__ mask,
                       uint32_t n_rows, uint32_t channels, int kind, float beta, float *__restrict__ partials)
{
    __shared__ float s_sum[EW_BLOCK / NSR_WAVE], s_cnt[EW_BLOCK / NSR_WAVE];
    const uint32_t n = n_rows * channels;
    float sum = 0.f, cnt = 0.f;
    for (uint32_t base = blockIdx.x * (EW_BLOCK * 8); base < n; base += gridDim.x * (EW_BLOCK * 8)) {
        float p[8], t[8];
        uint8_t m[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t i = base + k * EW_BLOCK + threadIdx.x;
            const bool in = i < n;
            const uint32_t j = in ? i : 0u;
            p[k] = pred[j];
            t[k] = target[j];
            m[k] = in ? mask[j / channels] : (uint8_t)0;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (m[k]) { sum += masked_loss_value(p[k] - t[k], kind, beta); cnt += 1.f; }
    }
    sum = wave_sum(sum);
    cnt = wave_sum(cnt);
    const int w = threadIdx.x / NSR_WAVE;
    if ((threadIdx.x & (NSR_WAVE - 1)) == 0) { s_sum[w] = sum; s_cnt[w] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float ts = 0.f, tc = 0.f;
        for (int k = 0; k < EW_BLOCK / NSR_WAVE; ++k) { ts += s_sum[k]; tc += s_cnt[k]; }
        partials[2 * blockIdx.x] = ts;
        partials[2 * blockIdx.x + 1] = tc;
    }
}

__global__ void __launch_bounds__(NSR_WAVE)
k_masked_loss_finish(const float *__restrict__ partials, uint32_t n_blocks, float *__restrict__ out)
{
    float ts = 0.f, tc = 0.f;
    for (uint32_t b = threadIdx.x; b < n_blocks; b += NSR_WAVE) { ts += partials[2 * b]; tc += partials[2 * b + 1]; }
    ts = wave_sum(ts);
    tc = wave_sum(tc);
    if (threadIdx.x == 0) {
        out[0] = tc > 0.f ? ts / tc : 0.f;  // (no valid row: 0 -- torch's mean over an empty selection is NaN)
        out[1] = tc;
    }
}

__global__ void __launch_bounds__(EW_BLOCK)
k_masked_loss_backward(const float *__restrict__ pred, const float *__restrict__ target, const uint8_t *__restrict__ mask,
                       uint32_t n_rows, uint32_t channels, int kind, float beta, const float *__restrict__ fwd_out,
                       const float *__restrict__ grad_out, float *__restrict__ d_pred)
{
    const uint32_t i = blockIdx.x * EW_BLOCK + threadIdx.x;
    if (i >= n_rows * channels) return;
    const float cnt = fwd_out[1];
    float g = 0.f;
    if (cnt > 0.f && mask[i / channels]) g = *grad_out * masked_loss_slope(pred[i] - target[i], kind, beta) / cnt;
    d_pred[i] = g;
}

}  // namespace

extern "C" int nsr_sh4_forward(const float *u, nsr_half *y, uint32_t n, uint32_t y_stride, void *stream)
{
    NSR_REQUIRE(y_stride >= 16, "nsr_sh4_forward: y_stride < 16");
    if (n == 0) return NSR_OK;
    NSR_REQUIRE(u && y, "nsr_sh4_forward: NULL pointer");
    hipLaunchKernelGGL(k_sh4, dim3(nsr_div_up(n, EW_BLOCK)), dim3(EW_BLOCK), 0, (hipStream_t)stream, u, (__half *)y, n,
                       y_stride);
    NSR_CHECK_LAUNCH("nsr_sh4_forward");
    return NSR_OK;
}

extern "C" int nsr_contract_to_unisphere(const float *x, float radius, int contraction, float *out, uint32_t n,
                                         void *stream)
{
    NSR_REQUIRE(contraction == NSR_CONTRACT_AABB || contraction == NSR_CONTRACT_UN_BOUNDED_SPHERE,
                "nsr_contract_to_unisphere: contraction type %d not implemented (reference raises too)", contraction);
    if (n == 0) return NSR_OK;
    NSR_REQUIRE(x && out, "nsr_contract_to_unisphere: NULL pointer");
    hipLaunchKernelGGL(k_contract_to_unisphere, dim3(nsr_div_up(n, EW_BLOCK)), dim3(EW_BLOCK), 0, (hipStream_t)stream,
                       x, radius, contraction, out, n);
    NSR_CHECK_LAUNCH("nsr_contract_to_unisphere");
    return NSR_OK;
}

extern "C" int nsr_density_activation_forward(const nsr_half *mlp_out, uint32_t stride, uint32_t n_feat, float bias,
                                              float *density, float *feature, uint32_t n, void *stream)
{
    if (n == 0) return NSR_OK;
    NSR_REQUIRE(mlp_out && density, "nsr_density_activation_forward: NULL pointer");
    hipLaunchKernelGGL(k_density_activation, dim3(nsr_div_up(n, EW_BLOCK)), dim3(EW_BLOCK), 0, (hipStream_t)stream,
                       (const __half *)mlp_out, stride, n_feat, bias, density, feature, n);
    NSR_CHECK_LAUNCH("nsr_density_activation_forward");
    return NSR_OK;
}

extern "C" int nsr_sample_positions(const float *rays_o, const float *rays_d, const int64_t *ray_indices,
                                    const float *t_starts, const float *t_ends, float *positions, float *dirs_out,
                                    uint32_t n, void *stream)
{
    if (n == 0) return NSR_OK;
    NSR_REQUIRE(rays_o && rays_d && ray_indices && t_starts && t_ends && positions,
                "nsr_sample_positions: NULL pointer");
    hipLaunchKernelGGL(k_sample_positions, dim3(nsr_div_up(n, EW_BLOCK)), dim3(EW_BLOCK), 0, (hipStream_t)stream,
                       rays_o, rays_d, ray_indices, t_starts, t_ends, positions, dirs_out, n);
    NSR_CHECK_LAUNCH("nsr_sample_positions");
    return NSR_OK;
}

extern "C" int nsr_neus_alpha_forward(const float *sdf, const float *normal, const float *dirs, const float *dists,
                                      const float *inv_s, float cos_anneal_ratio, float *alpha, uint32_t n,
                                      void *stream)
{
    if (n == 0) return NSR_OK;
    NSR_REQUIRE(sdf && normal && dirs && dists && inv_s && alpha, "nsr_neus_alpha_forward: NULL pointer");
    hipLaunchKernelGGL(k_neus_alpha_fwd, dim3(nsr_div_up(n, EW_BLOCK)), dim3(EW_BLOCK), 0, (hipStream_t)stream, sdf,
                       normal, dirs, dists, inv_s, cos_anneal_ratio, alpha, n);
    NSR_CHECK_LAUNCH("nsr_neus_alpha_forward");
    return NSR_OK;
}

extern "C" int nsr_neus_alpha_backward(const float *sdf, const float *normal, const float *dirs, const float *dists,
                                       const float *inv_s, float cos_anneal_ratio, const float *grad_alpha,
                                       float *grad_sdf, float *grad_normal, float *grad_inv_s, uint32_t n,
                                       void *stream)
{
    if (n == 0) return NSR_OK;
    NSR_REQUIRE(sdf && normal && dirs && dists && inv_s && grad_alpha, "nsr_neus_alpha_backward: NULL pointer");
    hipLaunchKernelGGL(k_neus_alpha_bwd, dim3(nsr_div_up(n, EW_BLOCK)), dim3(EW_BLOCK), 0, (hipStream_t)stream, sdf,
                       normal, dirs, dists, inv_s, cos_anneal_ratio, grad_alpha, grad_sdf, grad_normal, grad_inv_s, n);
    NSR_CHECK_LAUNCH("nsr_neus_alpha_backward");
    return NSR_OK;
}

extern "C" int nsr_adamw_step(float *params, float *grad, float *exp_avg, float *exp_avg_sq, nsr_half *shadow_half,
                              uint64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                              float bias_correction1, float bias_correction2, float grad_unscale, int zero_grad,
                              const float *hyper, uint64_t zero_first_n, void *stream)
{
    if (n == 0) return NSR_OK;
    if (zero_first_n == 0 || zero_first_n > n) zero_first_n = n;  // 0 = the whole gradient
    NSR_REQUIRE((zero_first_n & 3) == 0 || zero_first_n == n, "nsr_adamw_step: zero_first_n must be a multiple of 4");
    NSR_REQUIRE(params && grad && exp_avg && exp_avg_sq, "nsr_adamw_step: NULL pointer");
    NSR_REQUIRE((((uintptr_t)params | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
                "nsr_adamw_step: buffers must be 16-byte aligned");
    uint64_t blocks = (n / 4 + EW_BLOCK - 1) / EW_BLOCK + 1;
    if (blocks > 2048) blocks = 2048;  // grid-stride: ~8 blocks per CU
    hipLaunchKernelGGL(k_adamw, dim3((uint32_t)blocks), dim3(EW_BLOCK), 0, (hipStream_t)stream, params, grad, exp_avg,
                       exp_avg_sq, (__half *)shadow_half, n, lr, beta1, beta2, eps, weight_decay, bias_correction1,
                       bias_correction2, grad_unscale, zero_grad, hyper, zero_first_n,
                       nsr_guard.state ? nsr_guard.state + nsr_guard.parity : nullptr);
    NSR_CHECK_LAUNCH("nsr_adamw_step");
    return NSR_OK;
}

extern "C" int nsr_scale_to_half(const float *src, nsr_half *dst, uint64_t n, float scale, void *stream)
{
    if (n == 0) return NSR_OK;
    NSR_REQUIRE(src && dst, "nsr_scale_to_half: NULL pointer");
    NSR_REQUIRE((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "nsr_scale_to_half: buffers must be 16-byte aligned");
    uint64_t blocks = (n / 8 + EW_BLOCK - 1) / EW_BLOCK + 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_scale_to_half, dim3((uint32_t)blocks), dim3(EW_BLOCK), 0, (hipStream_t)stream, src,
                       (__half *)dst, n, scale);
    NSR_CHECK_LAUNCH("nsr_scale_to_half");
    return NSR_OK;
}

extern "C" int nsr_scale_from_half(const nsr_half *src, float *dst, uint64_t n, float scale, void *stream)
{
    if (n == 0) return NSR_OK;
    NSR_REQUIRE(src && dst, "nsr_scale_from_half: NULL pointer");
    NSR_REQUIRE((((uintptr_t)src | (uintptr_t)dst) & 15) == 0, "nsr_scale_from_half: buffers must be 16-byte aligned");
    uint64_t blocks = (n / 8 + EW_BLOCK - 1) / EW_BLOCK + 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_scale_from_half, dim3((uint32_t)blocks), dim3(EW_BLOCK), 0, (hipStream_t)stream,
                       (const __half *)src, dst, n, scale);
    NSR_CHECK_LAUNCH("nsr_scale_from_half");
    return NSR_OK;
}

extern "C" int nsr_adam_tick(int32_t *step, float *hyper, double base_lr, double beta1, double beta2, double gamma,
                             int32_t milestone0, int32_t milestone1, int32_t milestone2, void *stream)
{
    NSR_REQUIRE(step && hyper, "nsr_adam_tick: NULL pointer");
    hipLaunchKernelGGL(k_adam_tick, dim3(1), dim3(64), 0, (hipStream_t)stream, step, hyper, base_lr, beta1, beta2, gamma,
                       milestone0, milestone1, milestone2, nsr_guard.state, nsr_guard.parity);
    NSR_CHECK_LAUNCH("nsr_adam_tick");
    return NSR_OK;
}

extern "C" int nsr_adamw_step_scheduled_to(float *params_a, float *grad_a, float *exp_avg_a, float *exp_avg_sq_a,
                                           nsr_half *shadow_a, uint64_t n_a, uint64_t zero_first_n_a, float *params_b,
                                           float *grad_b, float *exp_avg_b, float *exp_avg_sq_b, nsr_half *shadow_b,
                                           uint64_t n_b, int32_t *step, float *hyper12, int32_t *step_out,
                                           float *hyper12_out, double base_lr, double beta1, double beta2, double gamma,
                                           int32_t milestone0, int32_t milestone1, int32_t milestone2, float eps,
                                           float weight_decay, float grad_unscale, int zero_grad, void *stream)
{
    NSR_REQUIRE(step && hyper12 && ((uintptr_t)hyper12 & 7u) == 0, "nsr_adamw_step_scheduled: step / hyper (8-byte aligned)");
    NSR_REQUIRE(step_out && hyper12_out && ((uintptr_t)hyper12_out & 7u) == 0 && (step_out == step) == (hyper12_out == hyper12),
                "nsr_adamw_step_scheduled: step_out / hyper_out (8-byte aligned; both in place or both elsewhere)");
    NSR_REQUIRE(n_a > 0 && params_a && grad_a && exp_avg_a && exp_avg_sq_a, "nsr_adamw_step_scheduled: NULL pointer");
    NSR_REQUIRE(n_b == 0 || (params_b && grad_b && exp_avg_b && exp_avg_sq_b), "nsr_adamw_step_scheduled: NULL pointer");
    NSR_REQUIRE((((uintptr_t)params_a | (uintptr_t)grad_a | (uintptr_t)exp_avg_a | (uintptr_t)exp_avg_sq_a |
                  (uintptr_t)params_b | (uintptr_t)grad_b | (uintptr_t)exp_avg_b | (uintptr_t)exp_avg_sq_b) & 15) == 0 &&
                    (((uintptr_t)shadow_a | (uintptr_t)shadow_b) & 7) == 0,
                "nsr_adamw_step_scheduled: buffers must be 16-byte aligned (fp16 shadow: 8)");
    NSR_REQUIRE(zero_first_n_a % 4 == 0, "nsr_adamw_step_scheduled: zero_first_n must be a multiple of 4");
    AdamSeg a{params_a, grad_a, exp_avg_a, exp_avg_sq_a, (__half *)shadow_a, n_a, zero_first_n_a ? zero_first_n_a : n_a};
    AdamSeg b{params_b, grad_b, exp_avg_b, exp_avg_sq_b, (__half *)shadow_b, n_b, n_b};
    uint64_t blocks = (n_a / 4 + EW_BLOCK - 1) / EW_BLOCK + 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_adamw_scheduled, dim3((uint32_t)blocks), dim3(EW_BLOCK), 0, (hipStream_t)stream, a, b, step,
                       hyper12, step_out, hyper12_out, base_lr, beta1, beta2, gamma, milestone0, milestone1, milestone2, (float)beta1,
                       (float)beta2, eps, weight_decay, grad_unscale, zero_grad, nsr_guard.state, nsr_guard.parity,
                       nsr_guard.scale0);
    NSR_CHECK_LAUNCH("nsr_adamw_step_scheduled");
    return NSR_OK;
}

NsrGuard nsr_guard = {nullptr, 0, 1.f};

// Register (state != NULL) / withdraw (NULL) the overflow guard for the launches QUEUED from now on by this process: the fused
// NeRF step's data-gradient kernel takes its loss scale from state[2] and reports non-finite gradients in state[parity], the
// table backward's fused AdamW and nsr_adamw_step_scheduled* skip the update when it is set, the latter also runs
// GradScaler.update() on the scale.  scale0: the constant the step descriptor's grad_scale holds.  Returns the parity the next
// step will use.
extern "C" int nsr_overflow_guard(int32_t *state, float scale0)
{
    nsr_guard.state = state;
    if (state) nsr_guard.scale0 = scale0 > 0.f ? scale0 : 1.f;
    return nsr_guard.parity;
}

extern "C" int nsr_adamw_step_scheduled(float *params_a, float *grad_a, float *exp_avg_a, float *exp_avg_sq_a,
                                        nsr_half *shadow_a, uint64_t n_a, uint64_t zero_first_n_a, float *params_b,
                                        float *grad_b, float *exp_avg_b, float *exp_avg_sq_b, nsr_half *shadow_b,
                                        uint64_t n_b, int32_t *step, float *hyper12, double base_lr, double beta1,
                                        double beta2, double gamma, int32_t milestone0, int32_t milestone1,
                                        int32_t milestone2, float eps, float weight_decay, float grad_unscale,
                                        int zero_grad, void *stream)
{
    return nsr_adamw_step_scheduled_to(params_a, grad_a, exp_avg_a, exp_avg_sq_a, shadow_a, n_a, zero_first_n_a, params_b,
                                       grad_b, exp_avg_b, exp_avg_sq_b, shadow_b, n_b, step, hyper12, step, hyper12, base_lr,
                                       beta1, beta2, gamma, milestone0, milestone1, milestone2, eps, weight_decay,
                                       grad_unscale, zero_grad, stream);
}

extern "C" int nsr_adamw_multi(const NsrAdamSegment *segments, uint32_t n_segments, float beta1, float beta2, float eps,
                               float weight_decay, float bias_correction1, float bias_correction2, int zero_grad,
                               int32_t *step_dev, float *hyper_dev, void *stream)
{
    NSR_REQUIRE((step_dev == nullptr) == (hyper_dev == nullptr), "nsr_adamw_multi: step_dev and hyper_dev come together");
    int32_t *guard = nsr_guard.state;
    if (step_dev) {  // (also with no live segment: the step count and the guard's scale move once per optimizer step)
        hipLaunchKernelGGL(k_adam_multi_tick, dim3(1), dim3(64), 0, (hipStream_t)stream, step_dev, hyper_dev, (double)beta1,
                           (double)beta2, guard, nsr_guard.parity);
        NSR_CHECK_LAUNCH("nsr_adamw_multi(tick)");
    }
    if (n_segments == 0) return NSR_OK;
    NSR_REQUIRE(segments && n_segments <= (uint32_t)ADAM_MULTI_MAX, "nsr_adamw_multi: 1..%d segments", ADAM_MULTI_MAX);
    AdamMulti t;
    uint32_t n_max = 0;
    for (uint32_t i = 0; i < n_segments; ++i) {
        const NsrAdamSegment &sg = segments[i];
        NSR_REQUIRE(sg.params && sg.grad && sg.exp_avg && sg.exp_avg_sq && sg.n < (1ull << 32),
                    "nsr_adamw_multi: segment %u has a NULL pointer or more than 2^32 - 1 elements", i);
        t.p[i] = sg.params; t.g[i] = sg.grad; t.m[i] = sg.exp_avg; t.v[i] = sg.exp_avg_sq;
        t.n[i] = (uint32_t)sg.n; t.lr[i] = sg.lr;
        n_max = sg.n > n_max ? (uint32_t)sg.n : n_max;
    }
    uint32_t bx = nsr_div_up(n_max, EW_BLOCK);
    if (bx > 64) bx = 64;
    if (bx == 0) bx = 1;
    hipLaunchKernelGGL(k_adamw_multi, dim3(bx, n_segments), dim3(EW_BLOCK), 0, (hipStream_t)stream, t, beta1, beta2, eps,
                       weight_decay, bias_correction1, bias_correction2, zero_grad, hyper_dev,
                       guard ? guard + nsr_guard.parity : nullptr);
    NSR_CHECK_LAUNCH("nsr_adamw_multi");
    return NSR_OK;
}

extern "C" uint32_t nsr_masked_loss_out_floats(void) { return 2 + 2 * ML_MAX_BLOCKS; }

extern "C" int nsr_masked_loss_forward(const float *pred, const float *target, const uint8_t *mask, uint32_t n_rows,
                                       uint32_t channels, int kind, float beta, float *out, void *stream)
{
    NSR_REQUIRE(out, "nsr_masked_loss_forward: NULL output");
    NSR_REQUIRE(kind >= 0 && kind <= 3, "nsr_masked_loss_forward: kind %d (0 smooth-L1, 1 MSE, 2 L1, 3 Huber)", kind);
    NSR_REQUIRE(channels >= 1 && (uint64_t)n_rows * channels < (1ull << 32), "nsr_masked_loss_forward: bad shape");
    NSR_REQUIRE(n_rows == 0 || (pred && target && mask), "nsr_masked_loss_forward: NULL pointer");
    uint32_t nb = nsr_div_up((uint64_t)n_rows * channels, EW_BLOCK * 8);
    if (nb > ML_MAX_BLOCKS) nb = ML_MAX_BLOCKS;
    if (nb > 0)
        hipLaunchKernelGGL(k_masked_loss_partials, dim3(nb), dim3(EW_BLOCK), 0, (hipStream_t)stream, pred, target, mask, n_rows,
                           channels, kind, beta, out + 2);
    hipLaunchKernelGGL(k_masked_loss_finish, dim3(1), dim3(NSR_WAVE), 0, (hipStream_t)stream, out + 2, nb, out);
    NSR_CHECK_LAUNCH("nsr_masked_loss_forward");
    return NSR_OK;
}

extern "C" int nsr_masked_loss_backward(const float *pred, const float *target, const uint8_t *mask, uint32_t n_rows,
                                        uint32_t channels, int kind, float beta, const float *forward_out,
                                        const float *grad_out, float *d_pred, void *stream)
{
    NSR_REQUIRE(kind >= 0 && kind <= 3, "nsr_masked_loss_backward: kind %d (0 smooth-L1, 1 MSE, 2 L1, 3 Huber)", kind);
    NSR_REQUIRE(channels >= 1 && (uint64_t)n_rows * channels < (1ull << 32), "nsr_masked_loss_backward: bad shape");
    if (n_rows == 0) return NSR_OK;
    NSR_REQUIRE(pred && target && mask && forward_out && grad_out && d_pred, "nsr_masked_loss_backward: NULL pointer");
    hipLaunchKernelGGL(k_masked_loss_backward, dim3(nsr_div_up((uint64_t)n_rows * channels, EW_BLOCK)), dim3(EW_BLOCK), 0,
                       (hipStream_t)stream, pred, target, mask, n_rows, channels, kind, beta, forward_out, grad_out, d_pred);
    NSR_CHECK_LAUNCH("nsr_masked_loss_backward");
    return NSR_OK;
}
