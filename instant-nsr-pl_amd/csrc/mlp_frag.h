// Register fragments of the fully fused 64-wide fp16 MLP on v_mfma_f32_16x16x32_f16 (csrc/mlp.hip explains the
// transposed, register-chained design; csrc/gridmlp.hip reuses it behind an in-register hash-grid encode).
#pragma once
#include "nsr_common.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MLP_BLOCK = 256;
constexpr int WAVES = MLP_BLOCK / 64;
constexpr int WIDTH = 64;

__device__ __forceinline__ f32x4 mfma32(half8 a, half8 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(half4 a, half4 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ int sigma(int kc, int g, int j) { return (2 * kc + (j >> 2)) * 16 + 4 * g + (j & 3); }

// A fragment of a row-major [rows, ld] half matrix W for output block ob, k-chunk kc.
//   natural: k = kc*32 + 8g + j        (first layer: B comes from memory in natural order)
//   permuted: k = sigma(kc, g, j)      (hidden layers: B is the previous layer's D registers)
__device__ __forceinline__ half8 load_a_natural(const _Float16 *__restrict__ W, int ld, int row, int kc, int g, int kmax)
{
    half8 a;
    const int k0 = kc * 32 + 8 * g;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = k0 + j;
        a[j] = (k < kmax) ? W[row * ld + k] : (_Float16)0;
    }
    return a;
}
__device__ __forceinline__ half8 load_a_sigma(const _Float16 *__restrict__ W, int ld, int row, int kc, int g)
{
    half8 a;
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = W[row * ld + sigma(kc, g, j)];
    return a;
}
// transposed fragments (dgrad): A[i][k] = W[k_index][i]
__device__ __forceinline__ half8 load_at_sigma(const _Float16 *__restrict__ W, int ld, int col, int kc, int g)
{
    half8 a;
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = W[sigma(kc, g, j) * ld + col];
    return a;
}
__device__ __forceinline__ half8 load_at_natural(const _Float16 *__restrict__ W, int ld, int col, int g, int kmax)
{
    half8 a;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * g + j;
        a[j] = (k < kmax) ? W[k * ld + col] : (_Float16)0;
    }
    return a;
}

// pack two D-layout accumulators (o-blocks 2kc, 2kc+1) into the next layer's B fragment
__device__ __forceinline__ half8 pack_b(const f32x4 &lo, const f32x4 &hi)
{
    half8 b;
#pragma unroll
    for (int r = 0; r < 4; ++r) { b[r] = (_Float16)lo[r]; b[4 + r] = (_Float16)hi[r]; }
    return b;
}

__device__ __forceinline__ void store_h4(__half *p, const f32x4 &v)
{
    __half2 h[2] = {__floats2half2_rn(v[0], v[1]), __floats2half2_rn(v[2], v[3])};
    *reinterpret_cast<uint2 *>(p) = *reinterpret_cast<uint2 *>(h);
}
__device__ __forceinline__ f32x4 load_h4(const __half *p)
{
    const uint2 raw = *reinterpret_cast<const uint2 *>(p);
    const __half2 a = *reinterpret_cast<const __half2 *>(&raw.x), b = *reinterpret_cast<const __half2 *>(&raw.y);
    f32x4 v = {__low2float(a), __high2float(a), __low2float(b), __high2float(b)};
    return v;
}

}  // namespace
