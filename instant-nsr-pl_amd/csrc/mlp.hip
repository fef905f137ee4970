// Fully fused 64-wide MLP on gfx950 MFMA (replaces tcnn.Network(FullyFusedMLP); reference call sites
// models/network_utils.py:181,209 ; weight layout documented at models/network_utils.py:142-173).
//
// Design (CDNA4-first; NOT tcnn's 128-thread / warp-WMMA / smem-weights tiling):
//   * everything is computed TRANSPOSED:  H^T = W . X^T  with  v_mfma_f32_16x16x32_f16.
//     A = 16 weight rows x 32 k, B = 32 k x 16 samples, D = 16 outputs x 16 samples.
//     In the D layout lane (n = lane&15, g = lane>>4) holds outputs {ob*16 + 4g + r, r<4} of ITS OWN
//     sample n.  The MFMA reduction index k is a dummy, so the next layer's B operand may enumerate
//     the hidden units in any order as long as A (the weights) uses the same order: we pick
//       sigma(kc, g, j) = (2kc + (j>>2))*16 + 4g + (j&3)
//     i.e. exactly the 8 values lane (n,g) already holds from o-blocks 2kc and 2kc+1.  Activations
//     therefore NEVER leave the lane's registers between layers: no LDS round trip, no shuffles.
//   * a wavefront owns all weights of the net in VGPRs (A fragments, permuted by sigma at load time:
//     72 VGPRs for 32->64->64->16) and streams 16-sample tiles through them (grid-stride).
//   * fp16 weights/activations, fp32 accumulation (tcnn accumulates in fp16).
//   * backward: a dgrad kernel (the same trick with W^T fragments; writes dX and the pre-activation gradients of every
//     layer column-blocked) and one wgrad kernel per weight matrix: dW = dY^T . X needs the SAMPLE index on the MFMA k
//     axis, so 32-sample tiles go through v_mfma_f32_16x16x32_f16 with K = the samples (fragments loaded straight from
//     the column-blocked buffers, 16-B loads); dW accumulates in fp32 VGPRs over all tiles of a wave, the waves of a
//     block meet through per-wave LDS regions and plain loads (NO LDS float atomics: ds_add_f32 retires 0.33 lane-ops
//     per clock on gfx950), one fp32 partial per block, summed by k_reduce_partials.
#include "nsr_common.h"
#include <stdlib.h>
#include "mlp_frag.h"

namespace {

// 8 consecutive input columns [c0, c0+8) of one sample row as fp16 (columns >= n_in are the constant 1)
// element (sample s, column c) of a LEVEL-MAJOR fp16 input [n_in/F][n][F]  (the fused path's encoding layout)
// column-blocked layout of an [n, cols] matrix for the wgrad kernels: per 32-sample tile a [cols][32] block
__device__ __forceinline__ uint64_t t32_off(uint32_t s, uint32_t col, uint32_t cols)
{
    return ((uint64_t)(s >> 5) * cols + col) * 32 + (s & 31);
}
__device__ __forceinline__ uint64_t lm_off(uint32_t s, int c, uint32_t n, uint32_t f) { return ((uint64_t)(c / f) * n + s) * f + c % f; }

__device__ __forceinline__ half8 load_x8(const void *__restrict__ x, bool x_f32, uint64_t row_off, int c0, int n_in,
                                         int in_pad, bool valid, uint32_t s = 0, uint32_t n = 0, uint32_t lmf = 0)
{
    half8 b;
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = (_Float16)0;
    if (!valid || c0 >= in_pad) return b;
    if (lmf) {
        const _Float16 *xh = reinterpret_cast<const _Float16 *>(x);
        if (lmf == 2 && c0 + 8 <= n_in) {  // four coalesced 4-B loads, one per level plane
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t v = *reinterpret_cast<const uint32_t *>(xh + ((uint64_t)(c0 / 2 + j) * n + s) * 2);
                const __half2 hv = *reinterpret_cast<const __half2 *>(&v);
                b[2 * j] = (_Float16)__low2float(hv);
                b[2 * j + 1] = (_Float16)__high2float(hv);
            }
            return b;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = c0 + j;
            if (c < n_in) b[j] = xh[lm_off(s, c, n, lmf)];
            else if (c < in_pad) b[j] = (_Float16)1;
        }
        return b;
    }
    if (!x_f32 && c0 + 8 <= n_in && ((row_off + c0) & 7) == 0) {
        b = *reinterpret_cast<const half8 *>(reinterpret_cast<const __half *>(x) + row_off + c0);
        return b;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = c0 + j;
        if (c < n_in)
            b[j] = x_f32 ? (_Float16) reinterpret_cast<const float *>(x)[row_off + c]
                         : reinterpret_cast<const _Float16 *>(x)[row_off + c];
        else if (c < in_pad)
            b[j] = (_Float16)1;
    }
    return b;
}

// 4 consecutive columns (D layout) as fp32
__device__ __forceinline__ f32x4 load_x4(const void *__restrict__ x, bool x_f32, uint64_t row_off, int c0, int n_in,
                                         int in_pad, bool valid, uint32_t s = 0, uint32_t n = 0, uint32_t lmf = 0)
{
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (!valid) return v;
    if (lmf) {
        const __half *xh = reinterpret_cast<const __half *>(x);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = c0 + r;
            if (c < n_in) v[r] = __half2float(xh[lm_off(s, c, n, lmf)]);
            else if (c < in_pad) v[r] = 1.f;
        }
        return v;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = c0 + r;
        if (c < n_in)
            v[r] = x_f32 ? reinterpret_cast<const float *>(x)[row_off + c]
                         : __half2float(reinterpret_cast<const __half *>(x)[row_off + c]);
        else if (c < in_pad)
            v[r] = 1.f;
    }
    return v;
}


// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int KIN /* in_pad/16 */, int NH>
__global__ void __launch_bounds__(MLP_BLOCK)
k_mlp_forward(const void *__restrict__ x, int x_f32, uint32_t x_stride, const __half *__restrict__ W_,
              __half *__restrict__ out, __half *__restrict__ acts, uint32_t n, uint32_t n_in, int out_act,
              uint32_t x_lmf, const int32_t *__restrict__ n_dev)
{
    const uint32_t n_live = live_count(n, n_dev);  // n stays the row stride of the level-major / per-layer arrays
    constexpr int IN_PAD = KIN * 16;
    constexpr int KC0 = (IN_PAD + 31) / 32;
    const int lane = threadIdx.x & 63, nl = lane & 15, g = lane >> 4;
    const uint32_t wave = (blockIdx.x * MLP_BLOCK + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * MLP_BLOCK) >> 6;
    const uint32_t n_tiles = (n_live + 15) / 16;
    const _Float16 *W = reinterpret_cast<const _Float16 *>(W_);

    // ---- weights -> registers ----
    half8 a0[4][KC0];
#pragma unroll
    for (int ob = 0; ob < 4; ++ob)
#pragma unroll
        for (int kc = 0; kc < KC0; ++kc) a0[ob][kc] = load_a_natural(W, IN_PAD, ob * 16 + nl, kc, g, IN_PAD);
    half8 ah[NH > 1 ? NH - 1 : 1][4][2];
#pragma unroll
    for (int h = 0; h < NH - 1; ++h) {
        const _Float16 *Wh = W + WIDTH * IN_PAD + h * WIDTH * WIDTH;
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) ah[h][ob][kc] = load_a_sigma(Wh, WIDTH, ob * 16 + nl, kc, g);
    }
    const _Float16 *Wl = W + WIDTH * IN_PAD + (NH - 1) * WIDTH * WIDTH;
    half8 al[2];
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) al[kc] = load_a_sigma(Wl, WIDTH, nl, kc, g);

    for (uint32_t tile = wave; tile < n_tiles; tile += n_waves) {
        const uint32_t s = tile * 16 + nl;
        const bool valid = s < n_live;
        const uint64_t row = (uint64_t)s * x_stride;
        f32x4 acc[4];
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) acc[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < KC0; ++kc) {
            const half8 b = load_x8(x, x_f32 != 0, row, kc * 32 + 8 * g, (int)n_in, IN_PAD, valid, s, n, x_lmf);
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) acc[ob] = mfma32(a0[ob][kc], b, acc[ob]);
        }
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            // ReLU, save, repack as next B
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[ob][r] = fmaxf(acc[ob][r], 0.f);
                if (acts && valid) store_h4(acts + ((uint64_t)h * n + s) * WIDTH + ob * 16 + 4 * g, acc[ob]);
            }
            const half8 b0 = pack_b(acc[0], acc[1]), b1 = pack_b(acc[2], acc[3]);
            if (h < NH - 1) {
#pragma unroll
                for (int ob = 0; ob < 4; ++ob) {
                    f32x4 c = {0.f, 0.f, 0.f, 0.f};
                    c = mfma32(ah[h][ob][0], b0, c);
                    acc[ob] = mfma32(ah[h][ob][1], b1, c);
                }
            } else {
                f32x4 c = {0.f, 0.f, 0.f, 0.f};
                c = mfma32(al[0], b0, c);
                c = mfma32(al[1], b1, c);
                if (out_act == NSR_ACT_SIGMOID) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) c[r] = 1.f / (1.f + __expf(-c[r]));
                }
                if (valid) store_h4(out + (uint64_t)s * 16 + 4 * g, c);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward = dgrad kernel + one wgrad kernel per weight matrix
//
// A single fused dgrad+wgrad kernel needed 252 VGPRs + accumulators (1 wave/SIMD, 171 blocks) and measured
// 23 % issue / 35 % dependency stalls / 41 % memory waits in a round-1 PMC pass (SQ_WAVE_CYCLES / SQ_WAIT_INST_ANY /
// SQ_INSTS_VALU; summarised in DESIGN.md section 4).  Split:
//   k_mlp_dgrad : same register-resident transposed chain as the forward (W^T fragments staged through LDS), one
//                 16-sample tile per wave, no accumulators -> many waves in flight.  Writes dX and, for the wgrad
//                 kernels, the pre-activation gradients of every layer (fp16, [n,64] rows; [n,16] for the output).
//   k_mlp_wgrad : dW[o][k] = sum_n G[n][o] A[n][k] as v_mfma_f32_16x16x32_f16 with the SAMPLE index on the MFMA k
//                 axis: 32-sample tiles are loaded row-major (64 B per lane, coalesced), transposed through a per-wave
//                 LDS tile (b16 scatter, b128 gather), accumulated in fp32 VGPRs over the wave's tiles, reduced over
//                 the block through per-wave LDS regions + plain loads (no LDS float atomics) and written as one partial
//                 per block (summed by k_reduce_partials).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lds_wave_sync()
{
    // order this wave's LDS writes before its LDS reads (tiles are wave-private).  Only the LDS counter is drained:
    // a workgroup-scope fence would also wait for every outstanding GLOBAL access (vmcnt(0)).
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

template <int KIN, int NH>
__global__ void __launch_bounds__(MLP_BLOCK)
k_mlp_dgrad(const void *__restrict__ dout, int dout_f32, uint32_t dout_stride, const float *__restrict__ dout_extra_col0,
            const __half *__restrict__ out, const __half *__restrict__ acts, const __half *__restrict__ W_,
            float *__restrict__ dx, uint32_t dx_stride, uint32_t dx_lm_features, __half *__restrict__ gpre,
            __half *__restrict__ gout, uint32_t ldn, uint32_t n, uint32_t n_in, uint32_t n_out, int out_act,
            float grad_scale, const int32_t *__restrict__ n_dev)
{
    const uint32_t n_live = live_count(n, n_dev);
    constexpr int IN_PAD = KIN * 16;
    constexpr int N_PARAMS = WIDTH * IN_PAD + (NH - 1) * WIDTH * WIDTH + 16 * WIDTH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, nl = lane & 15, g = lane >> 4;
    const uint32_t wave = (blockIdx.x * MLP_BLOCK + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * MLP_BLOCK) >> 6;
    const uint32_t n_tiles = (n_live + 15) / 16;
    {   // weights -> LDS with coalesced 16-B loads: the transposed fragments below are 2-byte strided gathers
        const _Float16 *Wg = reinterpret_cast<const _Float16 *>(W_);
        _Float16 *Wl_ = reinterpret_cast<_Float16 *>(smem);
        for (int k = threadIdx.x * 8; k < N_PARAMS; k += MLP_BLOCK * 8)
            *reinterpret_cast<uint4 *>(Wl_ + k) = *reinterpret_cast<const uint4 *>(Wg + k);
    }
    __syncthreads();
    const _Float16 *W = reinterpret_cast<const _Float16 *>(smem);
    const _Float16 *Wl = W + WIDTH * IN_PAD + (NH - 1) * WIDTH * WIDTH;  // [16, 64]
    half8 atl[4];  // dH^T (64 x n) = Wl^T (64 x 16) . dOut^T : K = 16 real (upper half of K = 32 is zero)
#pragma unroll
    for (int ib = 0; ib < 4; ++ib) atl[ib] = load_at_natural(Wl, WIDTH, ib * 16 + nl, g, 16);
    half8 ath[NH > 1 ? NH - 1 : 1][4][2];
#pragma unroll
    for (int h = 0; h < NH - 1; ++h) {
        const _Float16 *Wh = W + WIDTH * IN_PAD + h * WIDTH * WIDTH;
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) ath[h][ib][kc] = load_at_sigma(Wh, WIDTH, ib * 16 + nl, kc, g);
    }
    half8 at0[KIN][2];
    if (dx) {
#pragma unroll
        for (int ib = 0; ib < KIN; ++ib)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) at0[ib][kc] = load_at_sigma(W, IN_PAD, ib * 16 + nl, kc, g);
    }
    const float inv_scale = 1.f / grad_scale;

    for (uint32_t tile = wave; tile < n_tiles; tile += n_waves) {
        const uint32_t s = tile * 16 + nl;
        const bool valid = s < n_live;
        // dOut (D layout: cols 4g+r), output-activation derivative, scale
        f32x4 d_o = {0.f, 0.f, 0.f, 0.f};
        if (valid) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t c = 4 * g + r;
                if (c < n_out) {
                    const uint64_t off = (uint64_t)s * dout_stride + c;
                    float v = dout_f32 ? reinterpret_cast<const float *>(dout)[off]
                                       : __half2float(reinterpret_cast<const __half *>(dout)[off]);
                    if (c == 0 && dout_extra_col0) v += dout_extra_col0[s];
                    if (out_act == NSR_ACT_SIGMOID) {
                        const float o = __half2float(out[(uint64_t)s * 16 + c]);
                        v *= o * (1.f - o);
                    }
                    d_o[r] = v * grad_scale;
                }
            }
            if (gout) {  // column-blocked, 16 columns
#pragma unroll
                for (int r = 0; r < 4; ++r) gout[t32_off(s, 4 * g + r, 16)] = __float2half_rn(d_o[r]);
            }
        }
        half8 bo;  // natural-order B fragment of dOut^T, rebuilt from the D layout with shuffles
        {
            const int src_lo = nl + 16 * ((2 * g) & 3), src_hi = nl + 16 * ((2 * g + 1) & 3);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float lo = __shfl(d_o[r], src_lo, 64), hi = __shfl(d_o[r], src_hi, 64);
                bo[r] = (g < 2) ? (_Float16)lo : (_Float16)0;
                bo[4 + r] = (g < 2) ? (_Float16)hi : (_Float16)0;
            }
        }
        f32x4 dh[4];
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
            f32x4 c = {0.f, 0.f, 0.f, 0.f};
            dh[ib] = mfma32(atl[ib], bo, c);
        }
#pragma unroll
        for (int h = NH - 1; h >= 0; --h) {
            // ReLU backward with the saved post-activation of layer h; keep the result for the wgrad kernels
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) {
                const f32x4 hact = valid ? load_h4(acts + ((uint64_t)h * n + s) * WIDTH + ib * 16 + 4 * g)
                                         : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) dh[ib][r] = hact[r] > 0.f ? dh[ib][r] : 0.f;
                if (gpre && valid) {  // column-blocked, 64 columns per layer
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        gpre[(uint64_t)h * WIDTH * ldn + t32_off(s, ib * 16 + 4 * g + r, WIDTH)] =
                            __float2half_rn(dh[ib][r]);
                }
            }
            const half8 b0 = pack_b(dh[0], dh[1]), b1 = pack_b(dh[2], dh[3]);
            if (h > 0) {
#pragma unroll
                for (int ib = 0; ib < 4; ++ib) {
                    f32x4 c = {0.f, 0.f, 0.f, 0.f};
                    c = mfma32(ath[h > 0 ? h - 1 : 0][ib][0], b0, c);
                    dh[ib] = mfma32(ath[h > 0 ? h - 1 : 0][ib][1], b1, c);
                }
            } else if (dx) {
#pragma unroll
                for (int ib = 0; ib < KIN; ++ib) {
                    f32x4 c = {0.f, 0.f, 0.f, 0.f};
                    c = mfma32(at0[ib][0], b0, c);
                    c = mfma32(at0[ib][1], b1, c);
                    if (valid) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const uint32_t col = ib * 16 + 4 * g + r;
                            if (col < n_in) {
                                const uint64_t off = dx_lm_features
                                    ? ((uint64_t)(col / dx_lm_features) * n + s) * dx_lm_features + col % dx_lm_features
                                    : (uint64_t)s * dx_stride + col;
                                dx[off] = c[r] * inv_scale;
                            }
                        }
                    }
                }
            }
        }
    }
}

// ---- the data gradients of BOTH networks of the NeRF step in one kernel ------------------------------------------------------
// models/texture.py:24-26 feeds the colour network [16 geometry features | SH4(dir)]; the geometry features are the density
// network's outputs 0..15, so  dL/d(out_density) = dL/d(tex_in)[:, :16]  (+ dL/d logit on column 0): the colour network's input
// gradient IS the density network's output gradient, row by row.  In the transposed MFMA layout the first output block of the
// colour network's W0^T product leaves lane (n, g) with columns 4g .. 4g+3 of ITS sample -- exactly the D layout the density
// network's chain starts from.  So the 16 x fp32 d_feature row never goes through HBM (128 B written + 64 B read per sample by
// the two-kernel sequence), the SH half of d(tex_in) is never computed, and the step's chain has one launch (+ one event) less.
// Every value is formed by the same operations as in the two k_mlp_dgrad launches: dX, the saved pre-activation gradients and
// therefore the weight gradients are bit-identical (tests/test_gpu_mlp.py).
template <int NHC /* hidden layers, colour */, int NHD /* hidden layers, density */>
__global__ void __launch_bounds__(MLP_BLOCK, (NHC + NHD <= 3) ? 3 : 2)
k_mlp_dgrad_pair(const float *__restrict__ d_rgb /* [n,3] */, const float *__restrict__ d_logit /* [n] */,
                 const __half *__restrict__ out_c /* [n,16] sigmoid outputs */, const __half *__restrict__ acts_c,
                 const __half *__restrict__ Wc_, __half *__restrict__ gpre_c, __half *__restrict__ gout_c,
                 const __half *__restrict__ acts_d, const __half *__restrict__ Wd_, __half *__restrict__ gpre_d,
                 __half *__restrict__ gout_d, float *__restrict__ d_enc /* level-major [16][n][2] */, uint32_t ldn,
                 uint32_t n, float grad_scale, const int32_t *__restrict__ n_dev,
                 int32_t *__restrict__ guard /* NsrGuard state or NULL */, int parity)
{
    // overflow guard (nsr_common.h): the scale lives on the device, a non-finite d_enc -- every fp16 overflow upstream ends there,
    // the chain is dense -- sets this step's flag; the other step's flag is cleared here, where none of its readers is left
    if (guard) {
        grad_scale = __int_as_float(guard[2]);
        if (blockIdx.x == 0 && threadIdx.x == 0) guard[1 - parity] = 0;
    }
    bool bad = false;
    const uint32_t n_live = live_count(n, n_dev);
    constexpr int IN_PAD = 32;
    constexpr int NP_C = WIDTH * IN_PAD + (NHC - 1) * WIDTH * WIDTH + 16 * WIDTH;
    constexpr int NP_D = WIDTH * IN_PAD + (NHD - 1) * WIDTH * WIDTH + 16 * WIDTH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, nl = lane & 15, g = lane >> 4;
    const uint32_t wave = (blockIdx.x * MLP_BLOCK + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * MLP_BLOCK) >> 6;
    const uint32_t n_tiles = (n_live + 15) / 16;
    {
        _Float16 *Wl_ = reinterpret_cast<_Float16 *>(smem);
        const _Float16 *Wg = reinterpret_cast<const _Float16 *>(Wc_);
        for (int k = threadIdx.x * 8; k < NP_C; k += MLP_BLOCK * 8)
            *reinterpret_cast<uint4 *>(Wl_ + k) = *reinterpret_cast<const uint4 *>(Wg + k);
        Wg = reinterpret_cast<const _Float16 *>(Wd_);
        for (int k = threadIdx.x * 8; k < NP_D; k += MLP_BLOCK * 8)
            *reinterpret_cast<uint4 *>(Wl_ + NP_C + k) = *reinterpret_cast<const uint4 *>(Wg + k);
    }
    __syncthreads();
    const _Float16 *Wc = reinterpret_cast<const _Float16 *>(smem), *Wd = Wc + NP_C;
    // colour network: W_out^T, hidden W^T, and the rows 0..15 of W0^T (the only input columns whose gradient is needed)
    const _Float16 *Wcl = Wc + WIDTH * IN_PAD + (NHC - 1) * WIDTH * WIDTH;
    half8 atl_c[4];
#pragma unroll
    for (int ib = 0; ib < 4; ++ib) atl_c[ib] = load_at_natural(Wcl, WIDTH, ib * 16 + nl, g, 16);
    half8 ath_c[NHC > 1 ? NHC - 1 : 1][4][2];
#pragma unroll
    for (int h = 0; h < NHC - 1; ++h) {
        const _Float16 *Wh = Wc + WIDTH * IN_PAD + h * WIDTH * WIDTH;
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) ath_c[h][ib][kc] = load_at_sigma(Wh, WIDTH, ib * 16 + nl, kc, g);
    }
    half8 at0_c[2];
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) at0_c[kc] = load_at_sigma(Wc, IN_PAD, nl, kc, g);
    // density network
    const _Float16 *Wdl = Wd + WIDTH * IN_PAD + (NHD - 1) * WIDTH * WIDTH;
    half8 atl_d[4];
#pragma unroll
    for (int ib = 0; ib < 4; ++ib) atl_d[ib] = load_at_natural(Wdl, WIDTH, ib * 16 + nl, g, 16);
    half8 ath_d[NHD > 1 ? NHD - 1 : 1][4][2];
#pragma unroll
    for (int h = 0; h < NHD - 1; ++h) {
        const _Float16 *Wh = Wd + WIDTH * IN_PAD + h * WIDTH * WIDTH;
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) ath_d[h][ib][kc] = load_at_sigma(Wh, WIDTH, ib * 16 + nl, kc, g);
    }
    half8 at0_d[2][2];
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) at0_d[ib][kc] = load_at_sigma(Wd, IN_PAD, ib * 16 + nl, kc, g);
    const float inv_scale = 1.f / grad_scale;
    const int src_lo = nl + 16 * ((2 * g) & 3), src_hi = nl + 16 * ((2 * g + 1) & 3);

    for (uint32_t tile = wave; tile < n_tiles; tile += n_waves) {
        const uint32_t s = tile * 16 + nl;
        const bool valid = s < n_live;
        // ---------------- colour network: dOut = d_rgb * sigmoid' (columns 0..2), scaled -------------------------------------
        f32x4 d_o = {0.f, 0.f, 0.f, 0.f};
        if (valid) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t c = 4 * g + r;
                if (c < 3) {
                    float v = d_rgb[(uint64_t)s * 3 + c];
                    const float o = __half2float(out_c[(uint64_t)s * 16 + c]);
                    v *= o * (1.f - o);
                    d_o[r] = v * grad_scale;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) gout_c[t32_off(s, 4 * g + r, 16)] = __float2half_rn(d_o[r]);
        }
        half8 bo;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float lo = __shfl(d_o[r], src_lo, 64), hi = __shfl(d_o[r], src_hi, 64);
            bo[r] = (g < 2) ? (_Float16)lo : (_Float16)0;
            bo[4 + r] = (g < 2) ? (_Float16)hi : (_Float16)0;
        }
        f32x4 dh[4];
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
            f32x4 c = {0.f, 0.f, 0.f, 0.f};
            dh[ib] = mfma32(atl_c[ib], bo, c);
        }
        f32x4 dfeat = {0.f, 0.f, 0.f, 0.f};  // dL/d tex_in[:, 4g .. 4g+3] (unscaled), this lane's sample
#pragma unroll
        for (int h = NHC - 1; h >= 0; --h) {
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) {
                const f32x4 hact = valid ? load_h4(acts_c + ((uint64_t)h * n + s) * WIDTH + ib * 16 + 4 * g)
                                         : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) dh[ib][r] = hact[r] > 0.f ? dh[ib][r] : 0.f;
                if (valid) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        gpre_c[(uint64_t)h * WIDTH * ldn + t32_off(s, ib * 16 + 4 * g + r, WIDTH)] = __float2half_rn(dh[ib][r]);
                }
            }
            const half8 b0 = pack_b(dh[0], dh[1]), b1 = pack_b(dh[2], dh[3]);
            if (h > 0) {
#pragma unroll
                for (int ib = 0; ib < 4; ++ib) {
                    f32x4 c = {0.f, 0.f, 0.f, 0.f};
                    c = mfma32(ath_c[h > 0 ? h - 1 : 0][ib][0], b0, c);
                    dh[ib] = mfma32(ath_c[h > 0 ? h - 1 : 0][ib][1], b1, c);
                }
            } else {
                f32x4 c = {0.f, 0.f, 0.f, 0.f};
                c = mfma32(at0_c[0], b0, c);
                c = mfma32(at0_c[1], b1, c);
#pragma unroll
                for (int r = 0; r < 4; ++r) dfeat[r] = c[r] * inv_scale;  // (what the two-kernel path stores as d_tex)
            }
        }
        // ---------------- density network: dOut = d feature (+ d logit on column 0), no output activation -------------------
        if (valid) {
            if (g == 0) dfeat[0] += d_logit[s];
#pragma unroll
            for (int r = 0; r < 4; ++r) d_o[r] = dfeat[r] * grad_scale;
#pragma unroll
            for (int r = 0; r < 4; ++r) gout_d[t32_off(s, 4 * g + r, 16)] = __float2half_rn(d_o[r]);
        } else {
            d_o = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float lo = __shfl(d_o[r], src_lo, 64), hi = __shfl(d_o[r], src_hi, 64);
            bo[r] = (g < 2) ? (_Float16)lo : (_Float16)0;
            bo[4 + r] = (g < 2) ? (_Float16)hi : (_Float16)0;
        }
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
            f32x4 c = {0.f, 0.f, 0.f, 0.f};
            dh[ib] = mfma32(atl_d[ib], bo, c);
        }
#pragma unroll
        for (int h = NHD - 1; h >= 0; --h) {
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) {
                const f32x4 hact = valid ? load_h4(acts_d + ((uint64_t)h * n + s) * WIDTH + ib * 16 + 4 * g)
                                         : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) dh[ib][r] = hact[r] > 0.f ? dh[ib][r] : 0.f;
                if (valid) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        gpre_d[(uint64_t)h * WIDTH * ldn + t32_off(s, ib * 16 + 4 * g + r, WIDTH)] = __float2half_rn(dh[ib][r]);
                }
            }
            const half8 b0 = pack_b(dh[0], dh[1]), b1 = pack_b(dh[2], dh[3]);
            if (h > 0) {
#pragma unroll
                for (int ib = 0; ib < 4; ++ib) {
                    f32x4 c = {0.f, 0.f, 0.f, 0.f};
                    c = mfma32(ath_d[h > 0 ? h - 1 : 0][ib][0], b0, c);
                    dh[ib] = mfma32(ath_d[h > 0 ? h - 1 : 0][ib][1], b1, c);
                }
            } else {
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) {
                    f32x4 c = {0.f, 0.f, 0.f, 0.f};
                    c = mfma32(at0_d[ib][0], b0, c);
                    c = mfma32(at0_d[ib][1], b1, c);
                    if (valid) {  // level-major [16][n][2]: columns 4g+r of block ib are levels 8 ib + 2g and 8 ib + 2g + 1
                        const uint32_t lv = 8 * ib + 2 * g;
                        *reinterpret_cast<float2 *>(d_enc + ((uint64_t)lv * n + s) * 2) = make_float2(c[0] * inv_scale, c[1] * inv_scale);
                        *reinterpret_cast<float2 *>(d_enc + ((uint64_t)(lv + 1) * n + s) * 2) = make_float2(c[2] * inv_scale, c[3] * inv_scale);
                        bad |= !(fabsf(c[0]) + fabsf(c[1]) + fabsf(c[2]) + fabsf(c[3]) <= 3.4028234e38f);
                    }
                }
            }
        }
    }
    if (guard && __any(bad) && lane == 0) atomicOr(guard + parity, 1);
}

// ---- wgrad ------------------------------------------------------------------------------------------------------
// dW[o][k] = sum_n G[n][o] A[n][k] with the SAMPLE index on the MFMA k axis (v_mfma_f32_16x16x32_f16, 32 samples per
// instruction).  The fragment of lane (i = lane&15, g = lane>>4) is 8 CONSECUTIVE samples of ONE column: with the
// operands stored column-blocked (t32_off: per 32-sample tile a [cols][32] matrix, written that way by k_mlp_dgrad /
// k_mlp_forward) that is a single 16-B load and the whole wave reads 1 KB contiguous -- no LDS transpose, ~90 VGPRs,
// many waves in flight.  (Plain feature-major [cols][n] put the 16 columns of a fragment 2n bytes apart: 16 pages per
// load instruction, and the TLB misses made it slower than the LDS transpose.)  Row-major / level-major / fp32 operands (the first
// layer's input) fall back to 8 element loads.
//   a_kind 0: fp16 column-blocked (t32_off, ld = column count)   1: fp16 row-major [n, ld]   2: fp16 level-major [cols/F][n][F]
//          3: fp32 row-major [n, ld];  columns >= a_cols are the constant 1 (tcnn input padding)
template <int KIND>
__device__ __forceinline__ half8 wgrad_frag(const void *__restrict__ p, uint32_t ld, uint32_t lmf, uint32_t col, uint32_t n0,
                                            uint32_t n, uint32_t n_live, uint32_t n_cols)
{
    half8 f;
    if constexpr (KIND == 0) {  // ld = column count of the blocked buffer; the caller masks the ragged last tile
        const uint4 raw =
            *reinterpret_cast<const uint4 *>(reinterpret_cast<const _Float16 *>(p) + t32_off(n0, col, ld));
        f = *reinterpret_cast<const half8 *>(&raw);
    } else {
        const bool pad = col >= n_cols;
        const uint32_t c = pad ? 0 : col;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t s = n0 + j < n_live ? n0 + j : n_live - 1;  // clamped: unconditional loads, masked afterwards
            _Float16 v;
            if constexpr (KIND == 1) v = reinterpret_cast<const _Float16 *>(p)[(uint64_t)s * ld + c];
            else if constexpr (KIND == 2) v = reinterpret_cast<const _Float16 *>(p)[lm_off(s, (int)c, n, lmf)];
            else v = (_Float16) reinterpret_cast<const float *>(p)[(uint64_t)s * ld + c];
            f[j] = pad ? (_Float16)1 : v;
        }
    }
    return f;
}

__device__ __forceinline__ half8 wgrad_mask(half8 f, uint32_t n0, uint32_t n)
{
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (n0 + j >= n) f[j] = (_Float16)0;
    return f;
}

template <int OB /* output blocks of 16 */, int KB /* input blocks of 16 */, int KIND>
__global__ void __launch_bounds__(MLP_BLOCK)
k_mlp_wgrad(const __half *__restrict__ GT, const void *__restrict__ A, uint32_t a_ld, uint32_t a_lmf, uint32_t a_cols,
            float *__restrict__ partials, uint32_t partial_stride, uint32_t partial_offset, uint32_t n,
            const int32_t *__restrict__ n_dev)
{
    const uint32_t n_live = live_count(n, n_dev);  // rows; n is the row stride of level-major / row-major operands
    constexpr int OW = OB * 16, KW = KB * 16;
    static_assert((OB * KB) % 4 == 0, "block reduction assigns OB*KB/4 fragments to each thread");
    __shared__ f32x4 red[WAVES][OB * KB][64];  // per-wave accumulators in fragment order (LDS float atomics are ~4 clk/lane)
    const int lane = threadIdx.x & 63, nl = lane & 15, g = lane >> 4;
    const uint32_t wave = (blockIdx.x * MLP_BLOCK + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * MLP_BLOCK) >> 6;
    const uint32_t n_tiles = (n_live + 31) / 32;
    f32x4 acc[OB][KB];
#pragma unroll
    for (int a = 0; a < OB; ++a)
#pragma unroll
        for (int b = 0; b < KB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // software pipeline: the fragments of the wave's next tile are in flight while the current one is multiplied
    half8 ga[OB], ab[KB];
    auto load = [&](uint32_t tile, half8 *pg, half8 *pa) {
        const uint32_t n0 = tile * 32 + 8 * g;
#pragma unroll
        for (int a = 0; a < OB; ++a) pg[a] = wgrad_frag<0>(GT, OW, 0, a * 16 + nl, n0, n, n_live, OW);
#pragma unroll
        for (int b = 0; b < KB; ++b) pa[b] = wgrad_frag<KIND>(A, a_ld, a_lmf, b * 16 + nl, n0, n, n_live, a_cols);
    };
    uint32_t tile = wave;
    if (tile < n_tiles) load(tile, ga, ab);
    while (tile < n_tiles) {
        const uint32_t next = tile + n_waves;
        half8 ga2[OB], ab2[KB];
        if (next < n_tiles) load(next, ga2, ab2);
        if (tile * 32 + 32 > n_live) {  // ragged last tile: zero the samples past n on both operands
            const uint32_t n0 = tile * 32 + 8 * g;
#pragma unroll
            for (int a = 0; a < OB; ++a) ga[a] = wgrad_mask(ga[a], n0, n_live);
#pragma unroll
            for (int b = 0; b < KB; ++b) ab[b] = wgrad_mask(ab[b], n0, n_live);
        }
#pragma unroll
        for (int b = 0; b < KB; ++b)
#pragma unroll
            for (int a = 0; a < OB; ++a) acc[a][b] = mfma32(ga[a], ab[b], acc[a][b]);
#pragma unroll
        for (int a = 0; a < OB; ++a) ga[a] = ga2[a];
#pragma unroll
        for (int b = 0; b < KB; ++b) ab[b] = ab2[b];
        tile = next;
    }
    const int w = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < OB; ++a)
#pragma unroll
        for (int b = 0; b < KB; ++b) red[w][a * KB + b][lane] = acc[a][b];
    __syncthreads();
    // D layout of a dW block: lane (c = ln&15, g = ln>>4) of fragment (a,b) holds dW[a*16 + 4g + r][b*16 + c]
    float *dst = partials + (uint64_t)blockIdx.x * partial_stride + partial_offset;
#pragma unroll
    for (int i = 0; i < OB * KB / 4; ++i) {
        const int e = i * MLP_BLOCK + threadIdx.x, blk = e >> 6, ln = e & 63;
        f32x4 s = red[0][blk][ln];
#pragma unroll
        for (int k = 1; k < WAVES; ++k) s += red[k][blk][ln];
        const int a = blk / KB, b = blk % KB;
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(a * 16 + 4 * (ln >> 4) + r) * KW + b * 16 + (ln & 15)] = s[r];
    }
}

// grad[k] += inv_scale * sum_b partials[b][k].  2-D grid: x = 256-parameter column blocks, y = row segments; each
// block sums its rows with 8 independent accumulators and finishes with one fp32 atomic per parameter
// (n_params * RED_SEGS atomics in total: ~1e5, to distinct addresses).
constexpr int RED_SEGS = 16;
__global__ void __launch_bounds__(256)
k_reduce_partials(const float *__restrict__ partials, float *__restrict__ grad, uint32_t n_params, uint32_t n_blocks,
                  float inv_scale)
{
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_params) return;
    const uint32_t per = (n_blocks + RED_SEGS - 1) / RED_SEGS;
    const uint32_t b0 = blockIdx.y * per, b1 = min(n_blocks, b0 + per);
    // eight rows in flight per lane: the rows are n_params floats apart, every load is its own ~1 us round trip, and with
    // four in flight over 49 rows this 7 MB reduction took 21-34 us
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    uint32_t b = b0;
    for (; b + 8 <= b1; b += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] += partials[(uint64_t)(b + u) * n_params + k];
    }
    for (; b < b1; ++b) acc[0] += partials[(uint64_t)b * n_params + k];
    const float s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    if (b1 > b0) unsafeAtomicAdd(grad + k, s * inv_scale);
}

int check_mlp(const NsrMlpDesc *d, const char *who)
{
    NSR_REQUIRE(d != nullptr, "%s: desc is NULL", who);
    NSR_REQUIRE(d->in_pad % 16 == 0 && d->in_pad >= 16 && d->in_pad <= 64, "%s: in_pad=%u unsupported (16..64)", who,
                d->in_pad);
    NSR_REQUIRE(d->n_in >= 1 && d->n_in <= d->in_pad, "%s: n_in=%u > in_pad=%u", who, d->n_in, d->in_pad);
    NSR_REQUIRE(d->out_pad == 16, "%s: out_pad=%u unsupported (only 16)", who, d->out_pad);
    NSR_REQUIRE(d->n_out >= 1 && d->n_out <= d->out_pad, "%s: n_out=%u > out_pad", who, d->n_out);
    NSR_REQUIRE(d->n_hidden >= 1 && d->n_hidden <= 4, "%s: n_hidden_layers=%u unsupported (1..4)", who, d->n_hidden);
    NSR_REQUIRE(d->output_activation <= NSR_ACT_SIGMOID, "%s: unknown output activation %u", who, d->output_activation);
    return NSR_OK;
}

uint32_t n_params_of(const NsrMlpDesc *d) { return WIDTH * d->in_pad + (d->n_hidden - 1) * WIDTH * WIDTH + 16 * WIDTH; }

// wgrad: 32-sample tiles, ~2 tiles per wave, at most g_wgrad_cap blocks (one partial row of the whole network's parameters each:
// 28 KB for 32 -> 64 -> 64 -> 16, written by the wgrad kernels and read back by k_reduce_partials).  The cap is a run-time knob
// (nsr_mlp_wgrad_max_blocks, NSR_WGRAD_MAX_BLOCKS): with the step's join deferred to the next density MLP
// (nsr_nerf_wait_before_mlp) the weight-gradient kernels have ~170 us to finish, and FEWER blocks (each holds 64 KB of LDS)
// leave the table backward they run beside more of the chip.
constexpr uint32_t WGRAD_CAP_MAX = 512;
static uint32_t g_wgrad_cap = [] {
    const char *e = getenv("NSR_WGRAD_MAX_BLOCKS");
    const uint32_t v = e ? (uint32_t)atoi(e) : WGRAD_CAP_MAX;  // (the NeRF step lowers it around its own launches: csrc/step.hip)
    return v < 1 ? 1u : (v > WGRAD_CAP_MAX ? WGRAD_CAP_MAX : v);
}();
uint32_t bwd_blocks_capped(uint32_t n, uint32_t cap)
{
    const uint32_t n_tiles = (n + 31) / 32;
    uint32_t nb = (n_tiles + WAVES * 2 - 1) / (WAVES * 2);
    return nb < 1 ? 1 : (nb > cap ? cap : nb);
}
uint32_t bwd_blocks(uint32_t n) { return bwd_blocks_capped(n, g_wgrad_cap); }

// workspace layout (floats): [partials: nb * n_params][gpre^T: n_hidden * 64 * ldn halfs][gout^T: 16 * ldn halfs]
// (sized for the LARGEST cap: the knob may move between the allocation and the launches)
uint64_t bwd_ws_floats(const NsrMlpDesc *d, uint32_t n)
{
    const uint64_t ldn = (n + 31u) & ~31u;
    const uint64_t part = (uint64_t)bwd_blocks_capped(n, WGRAD_CAP_MAX) * n_params_of(d);
    return part + ((uint64_t)d->n_hidden * 64 * ldn + 16 * ldn) / 2 + 64;
}

}  // namespace

#define DISPATCH_MLP(KIN_, NH_, ...)                                                                     \
    switch ((KIN_) * 10 + (NH_)) {                                                                       \
    case 11: { constexpr int KIN = 1, NH = 1; __VA_ARGS__; } break;                                      \
    case 12: { constexpr int KIN = 1, NH = 2; __VA_ARGS__; } break;                                      \
    case 13: { constexpr int KIN = 1, NH = 3; __VA_ARGS__; } break;                                      \
    case 14: { constexpr int KIN = 1, NH = 4; __VA_ARGS__; } break;                                      \
    case 21: { constexpr int KIN = 2, NH = 1; __VA_ARGS__; } break;                                      \
    case 22: { constexpr int KIN = 2, NH = 2; __VA_ARGS__; } break;                                      \
    case 23: { constexpr int KIN = 2, NH = 3; __VA_ARGS__; } break;                                      \
    case 24: { constexpr int KIN = 2, NH = 4; __VA_ARGS__; } break;                                      \
    case 31: { constexpr int KIN = 3, NH = 1; __VA_ARGS__; } break;                                      \
    case 32: { constexpr int KIN = 3, NH = 2; __VA_ARGS__; } break;                                      \
    case 33: { constexpr int KIN = 3, NH = 3; __VA_ARGS__; } break;                                      \
    case 34: { constexpr int KIN = 3, NH = 4; __VA_ARGS__; } break;                                      \
    case 41: { constexpr int KIN = 4, NH = 1; __VA_ARGS__; } break;                                      \
    case 42: { constexpr int KIN = 4, NH = 2; __VA_ARGS__; } break;                                      \
    case 43: { constexpr int KIN = 4, NH = 3; __VA_ARGS__; } break;                                      \
    default: { constexpr int KIN = 4, NH = 4; __VA_ARGS__; } break;                                      \
    }

extern "C" int nsr_mlp_forward_ex(const void *x, int x_is_f32, uint32_t x_stride, uint32_t x_level_major_features,
                                  const nsr_half *weights, nsr_half *out, nsr_half *acts, uint32_t n,
                                  const NsrMlpDesc *desc, const int32_t *n_dev, void *stream);

extern "C" int nsr_mlp_forward(const void *x, int x_is_f32, uint32_t x_stride, const nsr_half *weights, nsr_half *out,
                               nsr_half *acts, uint32_t n, const NsrMlpDesc *desc, void *stream)
{
    return nsr_mlp_forward_ex(x, x_is_f32, x_stride, 0, weights, out, acts, n, desc, nullptr, stream);
}

static inline uint32_t mlp_ldn(uint32_t n) { return (n + 31u) & ~31u; }

extern "C" int nsr_mlp_forward_ex(const void *x, int x_is_f32, uint32_t x_stride, uint32_t x_level_major_features,
                                  const nsr_half *weights, nsr_half *out, nsr_half *acts, uint32_t n,
                                  const NsrMlpDesc *desc, const int32_t *n_dev, void *stream)
{
    if (int rc = check_mlp(desc, "nsr_mlp_forward")) return rc;
    if (n == 0) return NSR_OK;
    NSR_REQUIRE(x && weights && out, "nsr_mlp_forward: NULL pointer");
    NSR_REQUIRE(x_level_major_features || x_stride >= desc->n_in, "nsr_mlp_forward: x_stride < n_in");
    NSR_REQUIRE(!x_level_major_features || (!x_is_f32 && desc->n_in % x_level_major_features == 0),
                "nsr_mlp_forward: level-major input must be fp16 with n_in a multiple of the feature count");
    // the forward is latency-bound per tile (load -> MFMA chain -> store), so parallelism wins over weight reuse: one
    // 16-sample tile per wavefront until the chip is full (2048 blocks x 4 waves = 8 waves/SIMD), grid-stride beyond.
    // (measured at 8.8e4 samples: 8 tiles/wave 20 us -> 1 tile/wave)
    const uint32_t n_tiles = (n + 15) / 16;
    uint32_t blocks = (n_tiles + WAVES - 1) / WAVES;
    // a wave fetches all 14 KB of weights before its first tile: at most 512 workgroups, so that it streams >= 3 tiles per
    // fetch at the step's sizes (measured at 9.6e4 samples, 2 hidden layers: 2048 -> 22.4 us, 512 -> 14.5 us)
    if (blocks > 512) blocks = 512;
    DISPATCH_MLP(desc->in_pad / 16, desc->n_hidden,
                 hipLaunchKernelGGL((k_mlp_forward<KIN, NH>), dim3(blocks), dim3(MLP_BLOCK), 0, (hipStream_t)stream, x,
                                    x_is_f32, x_stride, (const __half *)weights, (__half *)out, (__half *)acts, n,
                                    desc->n_in, (int)desc->output_activation, x_level_major_features, n_dev));
    NSR_CHECK_LAUNCH("nsr_mlp_forward");
    return NSR_OK;
}

// at most `blocks` workgroups per weight-gradient launch (1 .. 512; 0 queries); returns the previous cap.  Data-gradient and
// weight-gradient halves of one backward must see the same value (they share the workspace layout).
extern "C" uint32_t nsr_mlp_wgrad_max_blocks(uint32_t blocks)
{
    const uint32_t old = g_wgrad_cap;
    if (blocks) g_wgrad_cap = blocks > WGRAD_CAP_MAX ? WGRAD_CAP_MAX : blocks;
    return old;
}

extern "C" uint64_t nsr_mlp_backward_workspace_floats(const NsrMlpDesc *desc, uint32_t n)
{
    if (!desc || check_mlp(desc, "nsr_mlp_backward_workspace_floats")) return 0;
    return bwd_ws_floats(desc, n);
}

template <int OB, int KB>
static void launch_wgrad(const __half *GT, const void *A, int a_kind, uint32_t a_ld, uint32_t a_lmf, uint32_t a_cols,
                         float *partials, uint32_t np, uint32_t off, uint32_t n, uint32_t nb, const int32_t *n_dev,
                         hipStream_t st)
{
#define NSR_WGRAD(KIND)                                                                                                \
    hipLaunchKernelGGL((k_mlp_wgrad<OB, KB, KIND>), dim3(nb), dim3(MLP_BLOCK), 0, st, GT, A, a_ld, a_lmf, a_cols,     \
                       partials, np, off, n, n_dev)
    switch (a_kind) {
    case 0: NSR_WGRAD(0); break;
    case 1: NSR_WGRAD(1); break;
    case 2: NSR_WGRAD(2); break;
    default: NSR_WGRAD(3); break;
    }
#undef NSR_WGRAD
}

// dgrad on `stream`; with a distinct `wgrad_stream` the weight-gradient kernels + their reduction are queued THERE behind
// the dgrad kernel (they read what it saved, nothing downstream of dx reads what they write): a caller that only needs
// dx to go on -- the next network's backward, the table backward -- keeps them off its critical path and joins
// `wgrad_stream` before the optimizer.
// phases: 1 = the dgrad kernel (also saves the pre-activation gradients when grad_weights is given), 2 = the weight-gradient
// kernels + their reduction (reading what a phase-1 call saved in `partials`), 3 = both
static int mlp_backward_impl(const void *dout, int dout_is_f32, uint32_t dout_stride, const float *dout_extra_col0,
                             const nsr_half *out, const void *x, int x_is_f32, uint32_t x_stride,
                             uint32_t x_level_major_features, const nsr_half *acts, const nsr_half *weights,
                             float *grad_weights, float *dx, uint32_t dx_stride, uint32_t dx_level_major_features,
                             float *partials, uint32_t n, float grad_scale, const NsrMlpDesc *desc, const int32_t *n_dev,
                             void *stream, void *wgrad_stream, int phases)
{
    if (int rc = check_mlp(desc, "nsr_mlp_backward")) return rc;
    if (n == 0) return NSR_OK;
    NSR_REQUIRE(dout && x && acts && weights, "nsr_mlp_backward: NULL pointer");
    NSR_REQUIRE(desc->output_activation == NSR_ACT_NONE || out, "nsr_mlp_backward: sigmoid backward needs `out`");
    NSR_REQUIRE(!grad_weights || partials, "nsr_mlp_backward: grad_weights needs the workspace");
    NSR_REQUIRE(!dx || dx_level_major_features || dx_stride >= desc->n_in, "nsr_mlp_backward: dx_stride < n_in");
    NSR_REQUIRE(!dx_level_major_features || desc->n_in % dx_level_major_features == 0,
                "nsr_mlp_backward: level-major dx needs n_in to be a multiple of the feature count");
    NSR_REQUIRE(!x_level_major_features || !x_is_f32, "nsr_mlp_backward: level-major x must be fp16");
    NSR_REQUIRE(grad_scale > 0.f, "nsr_mlp_backward: grad_scale must be > 0");
    hipStream_t st = (hipStream_t)stream;
    const uint32_t nb = bwd_blocks(n), np = n_params_of(desc), nh = desc->n_hidden, in_pad = desc->in_pad;
    const uint32_t ldn = mlp_ldn(n);
    __half *gpre = nullptr, *gout = nullptr;
    if (grad_weights) {
        gpre = reinterpret_cast<__half *>(partials + (uint64_t)nb * np);  // [nh][64][ldn]
        gout = gpre + (uint64_t)nh * 64 * ldn;                             // [16][ldn]
    }
    const uint32_t n_tiles = (n + 15) / 16;
    uint32_t blocks = (n_tiles + WAVES - 1) / WAVES;
    if (blocks > 2048) blocks = 2048;  // weights staged in LDS once per workgroup: the cap barely matters (256..2048 measured)
    if (phases & 1)
    DISPATCH_MLP(in_pad / 16, nh, {
        constexpr int NP = WIDTH * KIN * 16 + (NH - 1) * WIDTH * WIDTH + 16 * WIDTH;
        const size_t lds = NP * sizeof(_Float16);
        hipLaunchKernelGGL((k_mlp_dgrad<KIN, NH>), dim3(blocks), dim3(MLP_BLOCK), lds, st, dout, dout_is_f32, dout_stride,
                           dout_extra_col0, (const __half *)out, (const __half *)acts, (const __half *)weights, dx,
                           dx_stride, dx_level_major_features, gpre, gout, ldn, n, desc->n_in, desc->n_out,
                           (int)desc->output_activation, grad_scale, n_dev);
    });
    NSR_CHECK_LAUNCH("nsr_mlp_backward(dgrad)");
    if (!grad_weights || !(phases & 2)) return NSR_OK;
    if ((phases & 1) && wgrad_stream && wgrad_stream != stream) {
        static hipEvent_t ring[8] = {};
        static unsigned next = 0;
        hipEvent_t &ev = ring[next++ & 7u];
        if (!ev) NSR_REQUIRE(hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess, "nsr_mlp_backward: event");
        NSR_REQUIRE(hipEventRecord(ev, st) == hipSuccess &&
                        hipStreamWaitEvent((hipStream_t)wgrad_stream, ev, 0) == hipSuccess,
                    "nsr_mlp_backward: could not fork the weight-gradient stream");
        st = (hipStream_t)wgrad_stream;
    }
    const int x_kind = x_level_major_features ? 2 : (x_is_f32 ? 3 : 1);
    // first matrix W0 [64, in_pad]: G = gpre[0], A = x
    switch (in_pad / 16) {
    case 1: launch_wgrad<4, 1>(gpre, x, x_kind, x_stride, x_level_major_features, desc->n_in, partials, np, 0, n, nb, n_dev, st); break;
    case 2: launch_wgrad<4, 2>(gpre, x, x_kind, x_stride, x_level_major_features, desc->n_in, partials, np, 0, n, nb, n_dev, st); break;
    case 3: launch_wgrad<4, 3>(gpre, x, x_kind, x_stride, x_level_major_features, desc->n_in, partials, np, 0, n, nb, n_dev, st); break;
    default: launch_wgrad<4, 4>(gpre, x, x_kind, x_stride, x_level_major_features, desc->n_in, partials, np, 0, n, nb, n_dev, st); break;
    }
    // activations feeding matrix l >= 1: the forward's row-major [n,64] copy (element loads; measured as fast as a
    // column-blocked copy, so the forward does not write one)
    auto act_of = [&](uint32_t h, const void *&p, int &kind, uint32_t &ld) {
        p = (const __half *)acts + (uint64_t)h * n * 64; kind = 1; ld = 64;
    };
    for (uint32_t l = 1; l < nh; ++l) {  // hidden matrices W_l [64,64]: G = gpre[l], A = acts[l-1]
        const void *p; int kind; uint32_t ld;
        act_of(l - 1, p, kind, ld);
        launch_wgrad<4, 4>(gpre + (uint64_t)l * 64 * ldn, p, kind, ld, 0, 64, partials, np,
                           WIDTH * in_pad + (l - 1) * WIDTH * WIDTH, n, nb, n_dev, st);
    }
    {   // last matrix [16, 64]: G = gout, A = acts[nh-1]
        const void *p; int kind; uint32_t ld;
        act_of(nh - 1, p, kind, ld);
        launch_wgrad<1, 4>(gout, p, kind, ld, 0, 64, partials, np, WIDTH * in_pad + (nh - 1) * WIDTH * WIDTH, n, nb, n_dev, st);
    }
    NSR_CHECK_LAUNCH("nsr_mlp_backward(wgrad)");
    hipLaunchKernelGGL(k_reduce_partials, dim3(nsr_div_up(np, 256), RED_SEGS), dim3(256), 0, st, partials, grad_weights, np,
                       nb, 1.f / grad_scale);
    NSR_CHECK_LAUNCH("nsr_mlp_backward(reduce)");
    return NSR_OK;
}

// ---- both networks' data gradients in one launch (k_mlp_dgrad_pair) ----------------------------------------------------------
// 1 when the pair kernel covers these two networks: colour [16 features | 16 SH] -> 64 x (1..2) -> 3 sigmoid, density 32 -> 64 x
// (1..2) -> 16 linear outputs (the reference's nerf-blender shapes, models/texture.py:23-30 + models/geometry.py:122-130)
extern "C" int nsr_mlp_dgrad_pair_supported(const NsrMlpDesc *color, const NsrMlpDesc *density)
{
    if (!color || !density) return 0;
    return color->in_pad == 32 && color->n_in == 32 && color->n_out == 3 && color->out_pad == 16 &&
           color->output_activation == NSR_ACT_SIGMOID && color->n_hidden >= 1 && color->n_hidden <= 2 &&
           density->in_pad == 32 && density->n_in == 32 && density->n_out == 16 && density->out_pad == 16 &&
           density->output_activation == NSR_ACT_NONE && density->n_hidden >= 1 && density->n_hidden <= 2;
}

constexpr uint32_t g_dgrad_pair_max_blocks = 2048;
// d_rgb [n,3] fp32 and d_logit [n] fp32 in, d_enc level-major fp32 [16][n][2] out; the pre-activation gradients the
// weight-gradient kernels read are saved into the two networks' backward workspaces exactly where nsr_mlp_backward_phases(..., 1)
// puts them, so nsr_mlp_backward_phases(..., 2) follows unchanged.
extern "C" int nsr_mlp_dgrad_pair(const float *d_rgb, const float *d_logit, const nsr_half *out_color,
                                  const nsr_half *acts_color, const nsr_half *w_color, float *partials_color,
                                  const nsr_half *acts_density, const nsr_half *w_density, float *partials_density,
                                  float *d_enc_level_major, uint32_t n, float grad_scale, const NsrMlpDesc *color,
                                  const NsrMlpDesc *density, const int32_t *n_dev, void *stream)
{
    NSR_REQUIRE(nsr_mlp_dgrad_pair_supported(color, density), "nsr_mlp_dgrad_pair: network shapes not covered "
                "(colour 32 -> 64 x 1..2 -> 3 sigmoid, density 32 -> 64 x 1..2 -> 16)");
    if (n == 0) return NSR_OK;
    NSR_REQUIRE(d_rgb && d_logit && out_color && acts_color && w_color && partials_color && acts_density && w_density &&
                    partials_density && d_enc_level_major, "nsr_mlp_dgrad_pair: NULL pointer");
    NSR_REQUIRE(grad_scale > 0.f, "nsr_mlp_dgrad_pair: grad_scale must be > 0");
    const uint32_t nb = bwd_blocks(n), ldn = mlp_ldn(n);
    __half *gpre_c = reinterpret_cast<__half *>(partials_color + (uint64_t)nb * n_params_of(color));
    __half *gout_c = gpre_c + (uint64_t)color->n_hidden * 64 * ldn;
    __half *gpre_d = reinterpret_cast<__half *>(partials_density + (uint64_t)nb * n_params_of(density));
    __half *gout_d = gpre_d + (uint64_t)density->n_hidden * 64 * ldn;
    const uint32_t n_tiles = (n + 15) / 16;
    uint32_t blocks = (n_tiles + WAVES - 1) / WAVES;
    if (blocks > g_dgrad_pair_max_blocks) blocks = g_dgrad_pair_max_blocks;
    const size_t lds = (size_t)(n_params_of(color) + n_params_of(density)) * sizeof(_Float16);
#define NSR_PAIR(NHC, NHD)                                                                                               \
    NSR_LAUNCH_STOP((k_mlp_dgrad_pair<NHC, NHD>), dim3(blocks), dim3(MLP_BLOCK), lds, (hipStream_t)stream, d_rgb, d_logit, \
                       (const __half *)out_color, (const __half *)acts_color, (const __half *)w_color, gpre_c, gout_c,    \
                       (const __half *)acts_density, (const __half *)w_density, gpre_d, gout_d, d_enc_level_major, ldn, n, \
                       grad_scale, n_dev, guard, parity)
    // overflow guard (registered around a trainer's step, nsr_overflow_guard): this launch opens the step -- its flag is the
    // other one than the last step's; the optimizer launches queued behind it read nsr_guard.parity as it is left here
    int32_t *guard = nsr_guard.state;
    if (guard) nsr_guard.parity ^= 1;
    const int parity = nsr_guard.parity;
    switch (color->n_hidden * 10 + density->n_hidden) {
    case 11: NSR_PAIR(1, 1); break;
    case 12: NSR_PAIR(1, 2); break;
    case 21: NSR_PAIR(2, 1); break;
    default: NSR_PAIR(2, 2); break;
    }
#undef NSR_PAIR
    NSR_CHECK_LAUNCH("nsr_mlp_dgrad_pair");
    return NSR_OK;
}

NSR_INTERNAL int nsr_mlp_backward_split(const void *dout, int dout_is_f32, uint32_t dout_stride, const float *dout_extra_col0,
                                      const nsr_half *out, const void *x, int x_is_f32, uint32_t x_stride,
                                      uint32_t x_level_major_features, const nsr_half *acts, const nsr_half *weights,
                                      float *grad_weights, float *dx, uint32_t dx_stride,
                                      uint32_t dx_level_major_features, float *partials, uint32_t n, float grad_scale,
                                      const NsrMlpDesc *desc, const int32_t *n_dev, void *stream, void *wgrad_stream)
{
    return mlp_backward_impl(dout, dout_is_f32, dout_stride, dout_extra_col0, out, x, x_is_f32, x_stride, x_level_major_features,
                             acts, weights, grad_weights, dx, dx_stride, dx_level_major_features, partials, n, grad_scale, desc,
                             n_dev, stream, wgrad_stream, 3);
}

// the two halves of nsr_mlp_backward_split as separate calls: `phases` 1 = dgrad (+ what the weight-gradient kernels need,
// saved in `partials`), 2 = weight-gradient kernels + reduction on `stream`.  A step with two MLPs forks its helper stream
// ONCE behind the second dgrad instead of once per network (an event record costs the issuing stream ~8 us: csrc/step.hip)
extern "C" int nsr_mlp_backward_phases(const void *dout, int dout_is_f32, uint32_t dout_stride, const float *dout_extra_col0,
                                       const nsr_half *out, const void *x, int x_is_f32, uint32_t x_stride,
                                       uint32_t x_level_major_features, const nsr_half *acts, const nsr_half *weights,
                                       float *grad_weights, float *dx, uint32_t dx_stride,
                                       uint32_t dx_level_major_features, float *partials, uint32_t n, float grad_scale,
                                       const NsrMlpDesc *desc, const int32_t *n_dev, void *stream, int phases)
{
    NSR_REQUIRE(phases >= 1 && phases <= 3, "nsr_mlp_backward_phases: phases must be 1, 2 or 3");
    if (phases == 2 && n > 0) NSR_REQUIRE(x && acts && partials && grad_weights, "nsr_mlp_backward_phases: NULL pointer");
    return mlp_backward_impl(phases == 2 ? (dout ? dout : x) : dout, dout_is_f32, dout_stride, dout_extra_col0, out, x, x_is_f32,
                             x_stride, x_level_major_features, acts, weights, grad_weights, dx, dx_stride,
                             dx_level_major_features, partials, n, grad_scale, desc, n_dev, stream, nullptr, phases);
}

extern "C" int nsr_mlp_backward_ex(const void *dout, int dout_is_f32, uint32_t dout_stride, const float *dout_extra_col0,
                                   const nsr_half *out, const void *x, int x_is_f32, uint32_t x_stride,
                                   uint32_t x_level_major_features, const nsr_half *acts, const nsr_half *weights, float *grad_weights, float *dx, uint32_t dx_stride,
                                   uint32_t dx_level_major_features, float *partials, uint32_t n, float grad_scale,
                                   const NsrMlpDesc *desc, const int32_t *n_dev, void *stream)
{
    return nsr_mlp_backward_split(dout, dout_is_f32, dout_stride, dout_extra_col0, out, x, x_is_f32, x_stride,
                                  x_level_major_features, acts, weights, grad_weights, dx, dx_stride,
                                  dx_level_major_features, partials, n, grad_scale, desc, n_dev, stream, nullptr);
}

extern "C" int nsr_mlp_backward(const void *dout, int dout_is_f32, uint32_t dout_stride, const nsr_half *out,
                                const void *x, int x_is_f32, uint32_t x_stride, const nsr_half *acts,
                                const nsr_half *weights, float *grad_weights, float *dx, uint32_t dx_stride,
                                float *partials, uint32_t n, float grad_scale, const NsrMlpDesc *desc, void *stream)
{
    return nsr_mlp_backward_ex(dout, dout_is_f32, dout_stride, nullptr, out, x, x_is_f32, x_stride, 0, acts, weights, grad_weights, dx, dx_stride, 0, partials, n, grad_scale, desc, nullptr, stream);
}
