// Shared helpers for libnsr_hip.so (gfx950 only; 64-wide wavefronts assumed throughout).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/nsr_hip.h"

#define NSR_WAVE 64

void nsr_set_error(const char *fmt, ...);

#define NSR_REQUIRE(cond, ...)                \
    do {                                      \
        if (!(cond)) {                        \
            nsr_set_error(__VA_ARGS__);       \
            return NSR_ERR_INVALID;           \
        }                                     \
    } while (0)

#define NSR_CHECK_LAUNCH(name)                                               \
    do {                                                                     \
        hipError_t e_ = hipGetLastError();                                   \
        if (e_ != hipSuccess) {                                              \
            nsr_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return NSR_ERR_LAUNCH;                                           \
        }                                                                    \
    } while (0)

static inline uint32_t nsr_div_up(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// One-shot "stop event" of the NEXT launch made through NSR_LAUNCH_STOP on this thread: the event then rides on the kernel's
// own completion packet (hipExtLaunchKernelGGL) instead of costing the stream a packet of its own behind it -- measured 2.4 us
// less per fork on the step's chain (tools/event_cost.hip; tools/stop_event_order.hip: a waiter on another stream is ordered
// behind the kernel's completion).  The orchestration (csrc/step.hip) sets it right before the call whose kernel it belongs to
// and falls back to hipEventRecord when the callee did not consume it.
extern thread_local hipEvent_t nsr_next_stop_event;
#define NSR_LAUNCH_STOP(kernel, grid, block, lds, stream, ...)                                                       \
    do {                                                                                                             \
        hipEvent_t se_ = nsr_next_stop_event;                                                                        \
        nsr_next_stop_event = nullptr;                                                                               \
        if (se_) hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, nullptr, se_, 0, __VA_ARGS__);              \
        else hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                      \
    } while (0)

// Overflow guard of the fused fp16 training step (the reference trains under Lightning's `precision: 16`,
// configs/nerf-blender.yaml:103: GradScaler skips the optimizer step when a gradient is inf / NaN, halves the scale, and doubles
// it again after growth_interval clean steps).  state = int32[8] in device memory:
//   [0], [1] found-inf flags of alternating steps (the data-gradient kernel of step t sets [t & 1] and clears the other one;
//            every optimizer launch of step t reads [t & 1]: nothing ever resets a flag another kernel may still read)
//   [2] the scale (float bits)   [3] clean steps since the last change   [4] skipped steps   [5] growth interval
// Host-side registration only: the pointer is read when a launch is QUEUED (nsr_overflow_guard around a trainer's step), the
// kernels get it as an argument; nothing consults it afterwards.
struct NsrGuard { int32_t *state; int parity; float scale0; };
extern NsrGuard nsr_guard;

// Device-side row counts: an entry point that takes (n, n_dev) launches for the CAPACITY n and uses n for array
// strides; when n_dev != NULL only the first min(*n_dev, n) rows are live.  Lets a whole training step be queued
// without the host ever reading a sample count.
__device__ __forceinline__ uint32_t live_count(uint32_t n, const int32_t *__restrict__ n_dev)
{
    if (!n_dev) return n;
    const int32_t v = *n_dev;
    return v < 0 ? 0u : ((uint32_t)v < n ? (uint32_t)v : n);
}

// fp32 -> bf16, round to nearest even (NaN stays NaN); two values packed low | high << 16
__device__ __forceinline__ uint32_t nsr_to_bf16(float f)
{
    const uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7fc0u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ uint32_t nsr_pack_bf16x2(float lo, float hi) { return nsr_to_bf16(lo) | (nsr_to_bf16(hi) << 16); }

// ---- wave-level primitives (wave64) ------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// inclusive prefix sum across the 64 lanes of a wave
__device__ __forceinline__ float wave_incl_scan_add(float v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

__device__ __forceinline__ float wave_incl_scan_mul(float v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float t = __shfl_up(v, o, 64);
        if (lane >= o) v *= t;
    }
    return v;
}

// ---- AdamW arithmetic shared by every kernel that applies it (csrc/util.hip k_adamw*, the table backward's fused
// write-out in csrc/hashgrid.hip): torch.optim.AdamW semantics, the library is built with -ffp-contract=off, so the
// same inputs give the same bits wherever this is inlined ------------------------------------------------------------
// A non-finite gradient leaves the parameter and its moments untouched: the fused trainers run without a GradScaler
// (gradients leave the kernels in fp32), this is its "skip the step on overflow" at element granularity -- one overflowed
// sample must not turn a table entry into NaN for the rest of the run.  (torch.optim.AdamW would propagate the NaN.)
__device__ __forceinline__ void nsr_adamw_elem(float &p, float &m, float &v, float gr, float lr, float b1, float b2,
                                               float eps, float wd, float bc1, float bc2)
{
    if (!(fabsf(gr) <= 3.4028234e38f)) return;
    p *= (1.f - lr * wd);
    m = b1 * m + (1.f - b1) * gr;
    v = b2 * v + (1.f - b2) * gr * gr;
    const float denom = sqrtf(v) / sqrtf(bc2) + eps;
    p -= (lr / bc1) * (m / denom);
}

// (lr, bias corrections) of the optimizer step that FOLLOWS the `done` steps counted in *step -- MultiStepLR over base_lr,
// beta powers as running products kept in hyper[4..7] (doubles) by whoever advances the counter (k_adam_tick /
// k_adamw_scheduled).  Read-only; p1 / p2 return the new running products for the caller that publishes them.
__device__ __forceinline__ void nsr_adam_schedule(const int32_t *step, const float *hyper, double base_lr, double b1d,
                                                  double b2d, double gamma, int32_t m0, int32_t m1, int32_t m2,
                                                  float &lr, float &bc1, float &bc2, double &p1, double &p2)
{
    const int32_t done = *step, s = done + 1;
    const int k = (done >= m0) + (done >= m1) + (done >= m2);
    double scale = 1.0;
    for (int i = 0; i < k; ++i) scale *= gamma;
    const double *pw = reinterpret_cast<const double *>(hyper + 4);
    const int32_t pw_step = *reinterpret_cast<const int32_t *>(hyper + 3);
    if (pw_step == done && done > 0) { p1 = pw[0] * b1d; p2 = pw[1] * b2d; }
    else { p1 = pow(b1d, (double)s); p2 = pow(b2d, (double)s); }
    lr = (float)(base_lr * scale);
    bc1 = (float)(1.0 - p1);
    bc2 = (float)(1.0 - p2);
}

// ---- entry points shared between translation units of the library, NOT part of the C ABI (hidden visibility) ----------------
#define NSR_INTERNAL extern "C" __attribute__((visibility("hidden")))
// nsr_mlp_backward_ex with the weight-gradient kernels + reduction queued on `wgrad_stream` behind the dgrad kernel (NULL / ==
// stream: in line); the caller joins `wgrad_stream` before anything reads grad_weights  (csrc/mlp.hip, used by csrc/step.hip)
NSR_INTERNAL int nsr_mlp_backward_split(const void *dout, int dout_is_f32, uint32_t dout_stride, const float *dout_extra_col0,
                                        const nsr_half *out, const void *x, int x_is_f32, uint32_t x_stride,
                                        uint32_t x_level_major_features, const nsr_half *acts, const nsr_half *weights,
                                        float *grad_weights, float *dx, uint32_t dx_stride, uint32_t dx_level_major_features,
                                        float *partials, uint32_t n, float grad_scale, const NsrMlpDesc *desc,
                                        const int32_t *n_dev, void *stream, void *wgrad_stream);
// nsr_copy_ray_prefix_rows with per-array plane counts for level-major arrays (planes[q] row-arrays spaced src/dst_plane_bytes[q]
// apart) and, with tex_in, the texture network's input rows [n_kept, 32] = [first 16 halfs of tex_src[sample] | SH4(direction)]
NSR_INTERNAL int nsr_copy_ray_prefix_rows_ex(const int32_t *packed_old, const int32_t *packed_new, uint32_t n_arrays,
                                             const void *const *src, void *const *dst, const uint32_t *row_bytes,
                                             const uint32_t *planes, const uint64_t *src_plane_bytes,
                                             const uint64_t *dst_plane_bytes, const float *rays_d, float *dirs_out,
                                             int64_t *ray_indices_out, const nsr_half *tex_src, uint32_t tex_src_stride,
                                             nsr_half *tex_in, uint32_t n_rays, void *stream);
// nsr_composite_backward with dL/d weights[n] of further loss terms on the per-sample weights (grad_weights may be NULL)
NSR_INTERNAL int nsr_composite_backward_ex(const nsr_half *mlp_out, uint32_t stride, float density_bias, const float *t_starts,
                                           const float *t_ends, const nsr_half *rgb, uint32_t rgb_stride,
                                           const int32_t *packed_info, const float *background, const float *weights,
                                           const float *trans, const float *grad_comp_rgb, const float *grad_opacity,
                                           const float *grad_depth, const float *grad_weights, float *grad_rgb,
                                           float *grad_logit, uint32_t n_rays, void *stream);
