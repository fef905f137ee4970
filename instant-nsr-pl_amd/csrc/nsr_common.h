// Shared helpers for libnsr_hip.so (gfx950 only; 64-wide wavefronts assumed throughout).
#pragma once
#include <hip/hip_runtime.h>
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

// Device-side row counts: an entry point that takes (n, n_dev) launches for the CAPACITY n and uses n for array
// strides; when n_dev != NULL only the first min(*n_dev, n) rows are live.  Lets a whole training step be queued
// without the host ever reading a sample count.
__device__ __forceinline__ uint32_t live_count(uint32_t n, const int32_t *__restrict__ n_dev)
{
    if (!n_dev) return n;
    const int32_t v = *n_dev;
    return v < 0 ? 0u : ((uint32_t)v < n ? (uint32_t)v : n);
}

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
