// fp32 "VanillaMLP" (reference models/network_utils.py:95-139: nn.Linear stack WITH biases, 64 neurons, ReLU or
// Softplus(beta=100), optional weight norm folded by the caller) on gfx950 f32 MFMA -- the SDF network of the NeuS /
// neuralangelo configs (35 -> 64 -> 13, models/geometry.py:146-150) and the fp32 texture / background heads of
// configs/neus-dtu.yaml and configs/neuralangelo-dtu-wmask.yaml.  The reference runs these through cuBLAS GEMMs +
// elementwise kernels with autograd (and torch.autograd.grad(create_graph=True) for the analytic normal); here one
// wavefront streams 16-sample tiles through the whole net.
//
// Everything is computed TRANSPOSED with v_mfma_f32_16x16x4_f32 (exact fp32, 157 TFLOP/s = 2x the plain v_fma rate):
//   Z^T[neuron][sample] = W[neuron][feature] . X^T[feature][sample]
//   A operand: lane (c = lane&15, g = lane>>4) holds A[row c][k = g];  B: B[k = g][col c];  D reg r: D[4g + r][c].
// In the D layout a lane holds neurons {16 b + 4 g + r} of ITS sample c.  The MFMA reduction index is a dummy, so the
// next layer consumes D register r of block b as the B operand of "k-step (b, r)" and the weight fragment is loaded with
// the matching column 16 b + 4 g + r (one float4 per (b)): activations never leave registers, no shuffles, no LDS.
// The same trick runs the data gradient (W^T fragments).  The weight gradient needs the SAMPLE index on the k axis, i.e.
// both operands transposed: the tile's dZ and inputs go through a wave-private LDS tile ([sample][column]) and are read
// back in operand layout; dW accumulates in fp32 VGPRs over all tiles of the wave and is written once per wave as a
// partial (summed by k_vmlp_reduce: deterministic, no float atomics).  Bias gradients are register sums in D layout,
// reduced over the 16 sample lanes once at the end.
//
// SECOND (1 hidden layer, softplus): the analytic-normal protocol of models/geometry.py:177-180.  The forward kernel
// also emits g = d out[0] / d input = W0^T (s * u)  (s = sigmoid(100 z), u = last-layer row 0); the backward kernel takes
// P = dL/dg and adds the second-order terms  dW0 += (s*u) P^T,  du += s * (W0 P),  dz += 100 s (1-s) u (W0 P).
//
// Parameter blob (fp32): W0[64][in_pad] b0[64] | (W1[64][64] b1[64]) | Wl[16][64] bl[16]   (rows >= n_out of Wl/bl: 0).
#include "nsr_common.h"
#include <string.h>
#include <stdlib.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

#ifndef NSR_VMLP_RELOAD
#define NSR_VMLP_RELOAD 0
#endif
#ifndef NSR_VMLP_WAVES
#define NSR_VMLP_WAVES 1
#endif
constexpr bool VMLP_RELOAD_WEIGHTS = NSR_VMLP_RELOAD != 0;  // backward: re-load the weight fragments per tile (fewer registers)
constexpr int W = 64;        // hidden width
constexpr int LDT = 68;      // LDS tile row stride for 64-column tiles (floats)
constexpr int LDX = 52;      // LDS tile row stride for input tiles (<= 48 columns)
constexpr int LDO = 20;      // LDS tile row stride for the 16-column output-gradient tile

__device__ __forceinline__ void lds_wave_sync()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// Softplus(beta=100, threshold=20) and its derivative from ONE hardware exponential (v_exp_f32, ~1 ulp): with
// e = exp(-|t|), t = 100 z:  softplus = max(z, 0) + log(1 + e) / 100,  sigmoid(t) = t >= 0 ? 1 / (1 + e) : e / (1 + e).
// (libm's expf / log1pf cost ~60 VALU instructions per value and there are 16 values per lane and tile.)
// torch's thresholded branch (t > 20: identity, slope 1) needs NO special case: there e < 2.1e-9, 1 + e rounds to 1 in fp32,
// log(1) = 0 and 1 / 1 = 1 -- the formulas give z and 1 exactly.  Written with the branch, the compiler sank the three
// transcendentals under a divergent `if` per value: 16 saveexec / branch pairs per tile, each a scheduling barrier with its own
// exp -> log -> rcp latency chain (found in the ISA, round 6).
template <int ACT>
__device__ __forceinline__ float act_fwd(float z)
{
    if (ACT == 0) return fmaxf(z, 0.f);
    const float e = __expf(-fabsf(100.f * z));
    return fmaxf(z, 0.f) + __logf(1.f + e) * 0.01f;
}
// derivative of the activation w.r.t. its pre-activation (softplus: sigmoid(100 z))
template <int ACT>
__device__ __forceinline__ float act_bwd(float z)
{
    if (ACT == 0) return z > 0.f ? 1.f : 0.f;
    const float t = 100.f * z;
    const float e = __expf(-fabsf(t));
    const float r = __builtin_amdgcn_rcpf(1.f + e);  // (v_rcp_f32, 1 ulp; __frcp_rn is a ten-instruction IEEE division)
    return t >= 0.f ? r : e * r;
}

// activation and its derivative together (the backward kernels need both): ONE exponential
template <int ACT>
__device__ __forceinline__ void act_both(float z, float &a, float &s)
{
    if (ACT == 0) { a = fmaxf(z, 0.f); s = z > 0.f ? 1.f : 0.f; return; }
    const float t = 100.f * z;
    const float e = __expf(-fabsf(t));
    const float r = __builtin_amdgcn_rcpf(1.f + e);
    a = fmaxf(z, 0.f) + __logf(1.f + e) * 0.01f;
    s = t >= 0.f ? r : e * r;
}

struct Blob {
    const float *W0, *b0, *W1, *b1, *Wl, *bl;
};
template <int KS, int NH>
__device__ __forceinline__ Blob split_blob(const float *p)
{
    Blob b;
    b.W0 = p; p += W * KS * 4;
    b.b0 = p; p += W;
    b.W1 = b.b1 = nullptr;
    if (NH == 2) { b.W1 = p; p += W * W; b.b1 = p; p += W; }
    b.Wl = p; p += 16 * W;
    b.bl = p;
    return b;
}

// input feature k of sample s.  SDF_IN: [2 x01 - 1 (3) | hash encoding (fp16, row-major)] (CompositeEncoding with
// include_xyz, models/network_utils.py:75-76); otherwise fp32 rows
// enc_stride >= 0x80000000: level-major encoding [C/F][n][F] with F = enc_stride & 0xff and n = the row count (what the
// fused encode kernels write: a wave stores 64 x F consecutive halfs).  enc_stride & 0x40000000: tile-major
// [n / 16][C/F][16][F] -- the 16 rows of a tile contiguous, each level's 16 x F halfs contiguous inside it: a load
// instruction of this kernel (lanes = 16 rows x 4 consecutive columns) then touches 2-3 runs of 64 B instead of 16 rows
// All three layouts are  enc[row_base(s) + (c >> fs) * A + (c & fm)]  for column c: row-major A = 1, fs = 0 (stride in
// row_base); level-major A = n F; tile-major A = 16 F.  One formula, no divergent paths in the load loops.
struct EncLayout { uint32_t A, fs, fm; };
__device__ __forceinline__ EncLayout enc_layout(uint32_t enc_stride, uint32_t n)
{
    EncLayout e = {1u, 0u, 0u};
    if (enc_stride & 0xC0000000u) {
        const uint32_t F = enc_stride & 0xffu;
        e.fs = (uint32_t)__builtin_ctz(F); e.fm = F - 1u;
        e.A = (enc_stride & 0x80000000u) ? n * F : 16u * F;
    }
    return e;
}
__device__ __forceinline__ uint64_t enc_row_base(uint32_t enc_stride, uint64_t s, uint32_t n_in)
{
    if (!(enc_stride & 0xC0000000u)) return s * enc_stride;
    const uint32_t F = enc_stride & 0xffu;
    if (enc_stride & 0x80000000u) return s * F;
    return (s >> 4) * ((uint64_t)(n_in - 3u) * 16u) + (s & 15u) * F;  // (n_in - 3 = L F columns)
}

// Which input column lane (c, g) feeds into k-step kk of the first layer.  The MFMA reduction index is a dummy: any
// assignment works as long as the weight fragment uses the same one.  Natural: k = 4 kk + g.  PERMUTED (SDF input of 3 + 32
// columns, KS = 9): steps 0..7 carry encoding columns 8 g + kk -- lane g owns 8 CONSECUTIVE encoded features = 4 levels at
// F = 2 -- and step 8 carries x_g (g < 3; lane 3: the padding column).  A lane then fetches its share of a row with ONE
// 16-byte load (row-major) or four 4-byte loads that are 64-B runs across the 16 rows of the tile (tile- / level-major)
// instead of nine 2-byte loads.  (With the natural order the backward, which has no registers to keep nine loads in flight,
// serialised them: harmless for row-major, whose eight later loads hit the lines the first one fetched, but +15-36 % for
// the tile-major layout the encode kernels want to write.)
// (enc_only: a 32-input network fed by the encoding alone, x == NULL -- the NeRF++ background's density head in the grid
// refresh: columns 8 g + kk, step 8 carries the padding columns 32..35)
template <bool PERM>
__device__ __forceinline__ int kmap(int kk, int g, bool enc_only = false)
{
    if constexpr (PERM) {
        if (enc_only) return kk < 8 ? 8 * g + kk : 32 + g;
        return kk < 8 ? 3 + 8 * g + kk : (g < 3 ? g : 35);
    } else return 4 * kk + g;
}

// the lane's 8 encoded features + x_g for the permuted assignment.  enc_stride: row stride (multiple of 8), or
// 0x40000000 | 2 (tile-major), or 0x80000000 | 2 (level-major)
__device__ __forceinline__ void load_features_perm(const float *__restrict__ x, uint32_t x_stride,
                                                   const __half *__restrict__ enc, uint32_t enc_stride, uint64_t s, int g,
                                                   uint32_t n, float (&dst)[9])
{
    if (!(enc_stride & 0xC0000000u)) {
        const uint4 raw = *reinterpret_cast<const uint4 *>(enc + s * enc_stride + 8 * g);
        const uint32_t r[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const __half2 h = *reinterpret_cast<const __half2 *>(&r[j]);
            dst[2 * j] = __low2float(h); dst[2 * j + 1] = __high2float(h);
        }
    } else {
        const bool tile = (enc_stride & 0x40000000u) != 0u;
        // halfs between consecutive levels, and this lane's first level (4 g) of its row
        const uint64_t step = tile ? 32ull : 2ull * n;
        const __half *p = enc + (tile ? (s >> 4) * 512ull + (s & 15u) * 2u : s * 2ull) + (uint64_t)(4 * g) * step;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const __half2 h = *reinterpret_cast<const __half2 *>(p + j * step);
            dst[2 * j] = __low2float(h); dst[2 * j + 1] = __high2float(h);
        }
    }
    float xv = 0.f;
    if (x) xv = x[s * x_stride + (g < 3 ? g : 2)] * 2.f - 1.f;  // (wave-uniform branch; lane g = 3 re-reads column 2)
    dst[8] = g < 3 ? xv : 0.f;
}

template <bool SDF_IN>
__device__ __forceinline__ float load_feature(const float *__restrict__ x, uint32_t x_stride, const __half *__restrict__ enc,
                                              const EncLayout &el, uint64_t enc_base, uint32_t n_in, uint64_t s, uint32_t k)
{
    if (k >= n_in) return 0.f;
    if (SDF_IN) {
        if (k < 3) return x[s * x_stride + k] * 2.f - 1.f;
        const uint32_t c = k - 3;
        return __half2float(enc[enc_base + (c >> el.fs) * el.A + (c & el.fm)]);
    }
    return x[s * x_stride + k];
}

// chained-layer weight fragment: rows mb*16 + c, columns ib*16 + 4g .. +3 of a row-major [rows][64] matrix
__device__ __forceinline__ f32x4 load_chain(const float *__restrict__ M, int mb, int ib, int c, int g)
{
    return *reinterpret_cast<const f32x4 *>(M + (mb * 16 + c) * W + ib * 16 + 4 * g);
}

template <int KS, int NH, int ACT, bool SDF_IN, bool GRAD_IN>
__global__ void __launch_bounds__(64, (GRAD_IN && (KS >= 10 || !SDF_IN)) ? 1 : 2)  // (those variants need > 256 registers)
k_vmlp_forward(const float *__restrict__ blob, const float *__restrict__ x, uint32_t x_stride,
               const __half *__restrict__ enc, uint32_t enc_stride, uint32_t n_in, uint32_t n_out,
               float *__restrict__ out /* [n_full][16] */, float *__restrict__ out_col0 /* [n - n_full] */,
               float *__restrict__ g_in /* [n][KS*4] (GRAD_IN) */, uint32_t n, uint32_t n_full,
               const int32_t *__restrict__ n_dev)
{
    const uint32_t n_live = live_count(n, n_dev);
    const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
    const Blob B = split_blob<KS, NH>(blob);
    constexpr int IN_PAD = KS * 4;
    constexpr bool PERM = SDF_IN && KS == 9;  // [x (3) | 32 encoded features]: the permuted k assignment (kmap)
    constexpr int NB0 = (IN_PAD + 15) / 16;
    float wf0[4][KS];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) wf0[mb][kk] = B.W0[(mb * 16 + c) * IN_PAD + kmap<PERM>(kk, g, SDF_IN && !x)];
    f32x4 b0f[4], b1f[4], blf, wlf[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        b0f[mb] = *reinterpret_cast<const f32x4 *>(B.b0 + mb * 16 + 4 * g);
        if (NH == 2) b1f[mb] = *reinterpret_cast<const f32x4 *>(B.b1 + mb * 16 + 4 * g);
        wlf[mb] = load_chain(B.Wl, 0, mb, c, g);
    }
    blf = *reinterpret_cast<const f32x4 *>(B.bl + 4 * g);
    const uint32_t n_tiles = (n_live + 15) / 16;
    // software pipeline: the NEXT tile's inputs are requested before this tile's MFMA chain starts (one wave per SIMD: nothing
    // else hides the ~2 us of a global load)
    float xnext[KS];
    const EncLayout el = enc_layout(enc_stride, n);
    auto load_inputs = [&](uint32_t tile, float (&dst)[KS]) {
        const uint64_t sn = (uint64_t)tile * 16 + c;
        const bool ok = tile < n_tiles && sn < n_live;
        const uint64_t sc = ok ? sn : 0ull;  // (a row that exists: loads are unconditional, results selected)
        if constexpr (PERM) {
            load_features_perm(x, x_stride, enc, enc_stride, sc, g, n, dst);
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) dst[kk] = ok ? dst[kk] : 0.f;
        } else {
            // (plain fp32 rows: the conditional loads stay -- clamped + selected they cost the colour heads 12 %, measured)
            const uint64_t eb = SDF_IN ? enc_row_base(enc_stride, sc, n_in) : 0ull;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk)
                dst[kk] = ok ? load_feature<SDF_IN>(x, x_stride, enc, el, eb, n_in, sc, 4 * kk + g) : 0.f;
        }
    };
    load_inputs(blockIdx.x, xnext);
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t s = (uint64_t)tile * 16 + c;
        const bool valid = s < n_live;
        float xin[KS];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) xin[kk] = xnext[kk];
        load_inputs(tile + gridDim.x, xnext);
        f32x4 z[4], a[4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            z[mb] = b0f[mb];
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) z[mb] = mfma4(wf0[mb][kk], xin[kk], z[mb]);
#pragma unroll
            for (int r = 0; r < 4; ++r) a[mb][r] = act_fwd<ACT>(z[mb][r]);
        }
        if (NH == 2) {
            f32x4 z1[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                z1[mb] = b1f[mb];
#pragma unroll
                for (int ib = 0; ib < 4; ++ib) {
                    const f32x4 w = load_chain(B.W1, mb, ib, c, g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) z1[mb] = mfma4(w[r], a[ib][r], z1[mb]);
                }
            }
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) a[mb][r] = act_fwd<ACT>(z1[mb][r]);
        }
        if ((uint64_t)tile * 16 >= n_full) {
            // a tile of finite-difference taps: only out[0] = u . a + bl[0] is asked for -- 16 FMAs on the lane's own 16
            // neurons and a 4-lane reduction instead of the 16 MFMAs of the output layer
            float part = 0.f;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const f32x4 u = *reinterpret_cast<const f32x4 *>(B.Wl + mb * 16 + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) part = fmaf(u[r], a[mb][r], part);
            }
            part += __shfl_xor(part, 16, 64);
            part += __shfl_xor(part, 32, 64);
            if (valid && g == 0) out_col0[s - n_full] = part + B.bl[0];
        } else {
            f32x4 o = blf;
#pragma unroll
            for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                for (int r = 0; r < 4; ++r) o = mfma4(wlf[ib][r], a[ib][r], o);
            if (valid) {
                if (s < n_full) {
                    *reinterpret_cast<f32x4 *>(out + s * 16 + 4 * g) = o;
                } else if (g == 0) {
                    out_col0[s - n_full] = o[0];
                }
            }
        }
        if (GRAD_IN) {  // d out[0] / d input = W0^T (act'(z) * Wl[0][:])   (1 hidden layer)
            // the 48 W0^T fragments below are loop invariant: hoisted, they cost 48 registers and push the kernel past 256
            // (one wave per SIMD); re-read per tile from L1 (pointer made opaque) it fits two
            // (an opaque OFFSET, not an opaque pointer: the loads stay global_load -- behind an opaque pointer the compiler loses
            // the address space and emits flat_load, which waits on both memory counters)
            uint32_t opq = 0;
            asm volatile("" : "+s"(opq));
            const float *W0g = B.W0 + opq;
            f32x4 q[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const f32x4 u = *reinterpret_cast<const f32x4 *>(B.Wl + mb * 16 + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) q[mb][r] = act_bwd<ACT>(z[mb][r]) * u[r];
            }
#pragma unroll
            for (int fb = 0; fb < NB0; ++fb) {
                f32x4 gh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int col = fb * 16 + c;
                        // (clamped index + select: a conditional load costs a saveexec / branch pair each)
                        const float wv = W0g[(nb * 16 + 4 * g + r) * IN_PAD + (col < IN_PAD ? col : IN_PAD - 1)];
                        const float wt = col < IN_PAD ? wv : 0.f;
                        gh = mfma4(wt, q[nb][r], gh);
                    }
                if (valid) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int col = fb * 16 + 4 * g + r;
                        if (col < IN_PAD) g_in[s * IN_PAD + col] = gh[r];
                    }
                }
            }
        }
    }
}

// One wave per block: forward recompute + data gradient + weight gradient (+ second-order terms) of its tiles.
// Round 6 measured the alternative VERDICT r5 asked for -- a data-gradient kernel at two waves per SIMD that stores dz per layer
// as fp32 rows + a GEMM-shaped weight-gradient kernel with the rows on k -- and removed it: 735 vs 585 us over 7 x 2.6e5 points,
// 234 vs 177 us with second-order terms, 305 vs 283 us for the colour head (profiles/r06_vmlp_split_vs_fused.json); the dz round
// trip (256 B written + read per row and layer) costs more than the second wave buys.  What DID move this kernel (915 -> 585 us)
// was its instruction stream, read in the ISA: softplus written with torch's `t > 20` branch made the compiler sink the three
// transcendentals under a divergent if per value (16 saveexec / branch pairs per tile, each its own exp -> log -> rcp latency
// chain); `cond ? load : 0` and `if (cond) store` are a branch pair EACH (now: clamped index + select, stores through a sink
// word); __frcp_rn is a ten-instruction IEEE division (now v_rcp_f32); pointers made opaque with an empty asm lose their address
// space and load through flat_load (now: an opaque OFFSET); d_x column blocks nobody asked for were computed and stored.
// Register budget: 380-496 (arch + acc), one wave per SIMD, no spills -- after two fixes found in the ISA: the complete 64-bit
// per-column offsets of the d_x stores and of the final partial stores are functions of the lane only, so the compiler computed
// them BEFORE the tile loop and kept ~70 + ~100 registers alive across it (spilling 57-152 of them in the two-hidden-layer
// variants).  Now the d_x stores use twelve explicit 32-bit column parts + one 64-bit row base per tile, and the epilogue's
// lane coordinates are made opaque (empty asm) where its addresses are formed.  Measured against the round-2 kernel: colour
// heads (two hidden layers) 477 vs 576 us and 720 vs 898 us (C5 / C4), SDF network 1303 vs 1320 us (C5), 688 vs 715 us (C3).
// Measured alternative (round 3): a 128-thread workgroup with a "data" wave (forward / dgrad / d_x) feeding a "weight" wave
// (all dW accumulators) through the LDS tiles -- <= 248 registers, two waves per SIMD, no spills, all parity tests green --
// was SLOWER: C5 SDF backward 1.69 ms vs 1.32 ms, colour head 0.49 vs 0.50 ms, C3 step +4 %.  Two barriers per 16-sample
// tile and re-reading the weight fragments per tile cost more than the second wave hides; not adopted.
template <int KS, int NH, int ACT, bool SDF_IN, bool SECOND>
__global__ void __launch_bounds__(64, NSR_VMLP_WAVES)
k_vmlp_backward(const float *__restrict__ blob, const float *__restrict__ x, uint32_t x_stride,
                const __half *__restrict__ enc, uint32_t enc_stride, uint32_t n_in,
                const float *__restrict__ d_out /* [n_full][16] */, const float *__restrict__ d_out_col0,
                const float *__restrict__ p_in /* [n][KS*4], SECOND */, float *__restrict__ d_x, uint32_t dx_stride,
                uint32_t dx_first, uint32_t dx_count, uint32_t dx_lm_features, float *__restrict__ partials,
                uint32_t blob_floats, uint32_t n, uint32_t n_full, const int32_t *__restrict__ n_dev)
{
    static_assert(!SECOND || NH == 1, "second-order terms: one hidden layer");
    const uint32_t n_live = live_count(n, n_dev);
    const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
    const Blob B = split_blob<KS, NH>(blob);
    constexpr int IN_PAD = KS * 4;
    constexpr bool PERM = SDF_IN && KS == 9;  // [x (3) | 32 encoded features]: the permuted k assignment (kmap)
    constexpr int NB0 = (IN_PAD + 15) / 16;
    __shared__ __attribute__((aligned(16))) float T_a[16 * LDT];   // activations feeding a layer  [sample][neuron]
    __shared__ __attribute__((aligned(16))) float T_d[16 * LDT];   // pre-activation gradients     [sample][neuron]
    __shared__ __attribute__((aligned(16))) float T_x[16 * LDX];   // layer-0 inputs               [sample][feature]
    __shared__ __attribute__((aligned(16))) float T_o[16 * LDO];   // output gradients             [sample][output]
    for (int k = lane; k < 16 * LDX; k += 64) T_x[k] = 0.f;       // columns >= IN_PAD stay zero
    float wf0[4][KS];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) wf0[mb][kk] = B.W0[(mb * 16 + c) * IN_PAD + kmap<PERM>(kk, g)];
    f32x4 b0f[4], b1f[4], uf[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        b0f[mb] = *reinterpret_cast<const f32x4 *>(B.b0 + mb * 16 + 4 * g);
        if (NH == 2) b1f[mb] = *reinterpret_cast<const f32x4 *>(B.b1 + mb * 16 + 4 * g);
        uf[mb] = *reinterpret_cast<const f32x4 *>(B.Wl + mb * 16 + 4 * g);  // last-layer row 0 in D layout
    }
    // accumulators (D layout: rows 4g + r of the block, column c)
    f32x4 accW0[4][NB0], accW1[NH == 2 ? 4 : 1][4], accWl[4], db0[4], db1[4], du[4];
    float dbl = 0.f;  // lane (g,c): sum over its tiles of d_out[sample c][4kk + g] per kk -> kept per kk below
    f32x4 dblv = {0.f, 0.f, 0.f, 0.f};  // dblv[kk] = sum of d_out[.][4kk + g] seen by this lane
    (void)dbl;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
#pragma unroll
        for (int nb = 0; nb < NB0; ++nb) accW0[mb][nb] = zero4;
        accWl[mb] = zero4; db0[mb] = zero4; db1[mb] = zero4; du[mb] = zero4;
    }
    if constexpr (NH == 2) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) accW1[mb][nb] = zero4;
    }
    const uint32_t n_tiles = (n_live + 15) / 16;
    const uint32_t lm_shift = dx_lm_features ? 31u - (uint32_t)__clz((int)dx_lm_features) : 0u;
    // d_x column blocks.  The MFMA's row index is a dummy too: when only the 32 encoded columns of the [x | encoding] input are
    // asked for (what the table backward reads), two blocks starting at column 3 cover them -- 32 MFMAs instead of 48; and no
    // block beyond the last column asked for (the colour heads: 32 of 36 padded inputs -> two blocks, not three)
    const bool enc_only = PERM && dx_first >= 3u && dx_first + dx_count <= 35u;
    const int colbase = enc_only ? 3 : 0;
    const int nfb_want = (int)((dx_first + dx_count - colbase + 15u) / 16u);
    const int nfb = enc_only ? 2 : (nfb_want < NB0 ? nfb_want : NB0);
    // d_x addressing (see the store below): per-column offsets of this lane's NB0 x 4 output columns, 32 bits each
    uint32_t col_part[NB0 * 4], col_ok = 0u;
#pragma unroll
    for (int q = 0; q < NB0 * 4; ++q) {
        const uint32_t col = colbase + (q >> 2) * 16 + 4 * g + (q & 3);  // input feature of D register (fb = q >> 2, r = q & 3)
        const bool ok = (q >> 2) < nfb && col >= dx_first && col < dx_first + dx_count;
        const uint32_t k = ok ? col - dx_first : 0u;
        // level-major [k / F][n][F], F a power of two: ((k >> s) n + sample) << s | (k & (F - 1))
        col_part[q] = dx_lm_features ? ((((k >> lm_shift) * n) << lm_shift) | (k & (dx_lm_features - 1u))) : k;
        col_ok |= ok ? (1u << q) : 0u;
    }
    float *sink = partials + (uint64_t)blockIdx.x * blob_floats + lane;  // (this wave's partial slot: written in the epilogue only)
    // software pipeline: the NEXT tile's inputs (features, output gradient, P) are requested before this tile's MFMA chain
    float xnext[KS], pnext[KS];
    f32x4 donext;
    const EncLayout el = enc_layout(enc_stride, n);
    auto load_inputs = [&](uint32_t tile, float (&xd)[KS], float (&pd)[KS], f32x4 &dd) {
        const uint64_t sn = (uint64_t)tile * 16 + c;
        const bool ok = tile < n_tiles && sn < n_live;
        const uint64_t sc = ok ? sn : 0ull;  // (a row that exists: loads are unconditional, results selected)
        if constexpr (PERM) {
            load_features_perm(x, x_stride, enc, enc_stride, sc, g, n, xd);
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) xd[kk] = ok ? xd[kk] : 0.f;
        } else {
            // (plain fp32 rows: the conditional loads stay -- clamped + selected they cost the colour heads 12 %, measured)
            const uint64_t eb = SDF_IN ? enc_row_base(enc_stride, sc, n_in) : 0ull;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) xd[kk] = ok ? load_feature<SDF_IN>(x, x_stride, enc, el, eb, n_in, sc, 4 * kk + g) : 0.f;
        }
        if (SECOND) {
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) pd[kk] = ok ? p_in[sn * IN_PAD + kmap<PERM>(kk, g)] : 0.f;
        }
        dd = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ok) {
            if (sn < n_full) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) dd[kk] = d_out[sn * 16 + 4 * kk + g];
            } else {
                dd[3] = d_out_col0[sn - n_full];   // every lane of the sample: the tap fast path reads it from slot 3
                if (g == 0) dd[0] = dd[3];         // B-layout slot of output column 0
            }
        }
    };
    load_inputs(blockIdx.x, xnext, pnext, donext);
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t s = (uint64_t)tile * 16 + c;
        const bool valid = s < n_live;
        // two hidden layers: ~250 loop-invariant weight fragments would be hoisted into registers and spilled; re-load the
        // (L1-resident, 37 KB) matrices per tile instead by hiding the pointers' loop invariance from the compiler
        const float *W0t = B.W0, *W1t = B.W1, *Wlt = B.Wl;
        if constexpr (NH == 2 || VMLP_RELOAD_WEIGHTS) {
            // (opaque POINTERS here, flat_load and all: with an opaque offset the two-hidden-layer variants measured 12 % slower --
            // the compiler keeps more of the address arithmetic live across the tile)
            asm volatile("" : "+s"(W0t), "+s"(W1t), "+s"(Wlt));
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) wf0[mb][kk] = W0t[(mb * 16 + c) * IN_PAD + kmap<PERM>(kk, g)];
        }
        // ---- forward recompute ---------------------------------------------------------------------------------
        float xin[KS], pb[KS];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) { xin[kk] = xnext[kk]; pb[kk] = SECOND ? pnext[kk] : 0.f; }
        f32x4 dob = donext;
        load_inputs(tile + gridDim.x, xnext, pnext, donext);
        // ReLU: act'(z) = (a > 0), so neither the pre-activations nor a separate derivative array stay live (the two-hidden-
        // layer colour heads otherwise spill); softplus keeps s0 = sigmoid(100 z0), evaluated once (the curvature term is
        // 100 s0 (1 - s0))
        f32x4 a0[4], s0[ACT == 1 ? 4 : 1], a1[4], s1[(NH == 2 && ACT == 1) ? 4 : 1];
#define NSR_S0(mb, r) (ACT == 1 ? s0[ACT == 1 ? (mb) : 0][r] : (a0[mb][r] > 0.f ? 1.f : 0.f))
#define NSR_S1(mb, r) (ACT == 1 ? s1[(NH == 2 && ACT == 1) ? (mb) : 0][r] : (a1[mb][r] > 0.f ? 1.f : 0.f))
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            f32x4 z = b0f[mb];
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) z = mfma4(wf0[mb][kk], xin[kk], z);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float av, sv;
                act_both<ACT>(z[r], av, sv);
                a0[mb][r] = av;
                if constexpr (ACT == 1) s0[mb][r] = sv;
            }
        }
        if constexpr (NH == 2) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                f32x4 z = b1f[mb];
#pragma unroll
                for (int ib = 0; ib < 4; ++ib) {
                    const f32x4 w = load_chain(W1t, mb, ib, c, g);
#pragma unroll
                    for (int r = 0; r < 4; ++r) z = mfma4(w[r], a0[ib][r], z);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float av, sv;
                    act_both<ACT>(z[r], av, sv);
                    a1[mb][r] = av;
                    if constexpr (ACT == 1) s1[mb][r] = sv;
                }
            }
        }
        // ---- output gradient in B layout: lane (g,c) holds d_out[sample c][4kk + g] -------------------------------
        const bool tap_tile = (uint64_t)tile * 16 >= n_full;  // finite-difference taps: only d out[0] is non-zero
        const float d_tap = tap_tile ? dob[3] : 0.f;
        if (s >= n_full) dob[3] = 0.f;  // (slot 3 carried the tap's scalar; as an output-gradient slot it is column 12 + g: zero)
        dblv += dob;
        // dA_last^T = Wl^T . dOut^T   (A: Wl[4kk + g][fb*16 + c])
        f32x4 dz_last[4];
        if (tap_tile) {
            // rank-1 output gradient: dz = act'(z) * Wl[0][:] * d, and dWl row 0 += d * a -- per-lane products on the D
            // layout (the row-0 sum joins accWl at the end like the second-order du does): no MFMA, no LDS round trip
#pragma unroll
            for (int fb = 0; fb < 4; ++fb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dz_last[fb][r] = d_tap * uf[fb][r] * ((NH == 2) ? NSR_S1(fb, r) : NSR_S0(fb, r));
                    du[fb][r] += d_tap * ((NH == 2) ? a1[fb][r] : a0[fb][r]);
                }
        } else {
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
            f32x4 acc = zero4;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc = mfma4(Wlt[(4 * kk + g) * W + fb * 16 + c], dob[kk], acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) dz_last[fb][r] = acc[r] * ((NH == 2) ? NSR_S1(fb, r) : NSR_S0(fb, r));
        }
        // ---- last-layer weight gradient: dWl[o][j] += sum_s dOut[o][s] a_last[j][s] ------------------------------
        lds_wave_sync();  // previous tile's readers are done
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
            *reinterpret_cast<f32x4 *>(&T_a[c * LDT + mb * 16 + 4 * g]) = (NH == 2) ? a1[mb] : a0[mb];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) T_o[c * LDO + 4 * kk + g] = dob[kk];
        lds_wave_sync();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const float ao = T_o[(4 * kk + g) * LDO + c];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) accWl[nb] = mfma4(ao, T_a[(4 * kk + g) * LDT + nb * 16 + c], accWl[nb]);
        }
        }  // !tap_tile
        f32x4 dz0[4];
        if constexpr (NH == 2) {
            // dW1[i][j] += sum_s dz1[i][s] a0[j][s] ;  db1 ;  dA0^T = W1^T dz1^T ; dz0 = dA0 * act'(z0)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) db1[mb] += dz_last[mb];
            lds_wave_sync();
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                *reinterpret_cast<f32x4 *>(&T_d[c * LDT + mb * 16 + 4 * g]) = dz_last[mb];
                *reinterpret_cast<f32x4 *>(&T_a[c * LDT + mb * 16 + 4 * g]) = a0[mb];
            }
            lds_wave_sync();
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                float bj[4];
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) bj[nb] = T_a[(4 * kk + g) * LDT + nb * 16 + c];
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    const float ai = T_d[(4 * kk + g) * LDT + mb * 16 + c];
#pragma unroll
                    for (int nb = 0; nb < 4; ++nb) accW1[mb][nb] = mfma4(ai, bj[nb], accW1[mb][nb]);
                }
            }
#pragma unroll
            for (int fb = 0; fb < 4; ++fb) {
                f32x4 acc = zero4;
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc = mfma4(W1t[(nb * 16 + 4 * g + r) * W + fb * 16 + c], dz_last[nb][r], acc);
#pragma unroll
                for (int r = 0; r < 4; ++r) dz0[fb][r] = acc[r] * NSR_S0(fb, r);
            }
        } else {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) dz0[mb] = dz_last[mb];
        }
        // ---- second-order terms of the analytic normal -----------------------------------------------------------
        if (SECOND) {
            f32x4 q[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                f32x4 dq = zero4;
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) dq = mfma4(wf0[mb][kk], pb[kk], dq);  // (W0 P)^T
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sg = NSR_S0(mb, r);
                    q[mb][r] = sg * uf[mb][r];
                    du[mb][r] += sg * dq[r];
                    // d/dz of act'(z): softplus -> 100 s (1 - s) (0 on torch's linear branch); relu -> 0
                    const float curv = ACT == 1 ? 100.f * sg * (1.f - sg) : 0.f;  // (0 exactly where torch takes its linear branch: s = 1)
                    dz0[mb][r] += curv * uf[mb][r] * dq[r];
                }
            }
            // dW0 += q P^T
            lds_wave_sync();
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) *reinterpret_cast<f32x4 *>(&T_d[c * LDT + mb * 16 + 4 * g]) = q[mb];
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) T_x[c * LDX + kmap<PERM>(kk, g)] = pb[kk];
            lds_wave_sync();
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                float bj[NB0];
#pragma unroll
                for (int nb = 0; nb < NB0; ++nb) bj[nb] = T_x[(4 * kk + g) * LDX + nb * 16 + c];
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    const float ai = T_d[(4 * kk + g) * LDT + mb * 16 + c];
#pragma unroll
                    for (int nb = 0; nb < NB0; ++nb) accW0[mb][nb] = mfma4(ai, bj[nb], accW0[mb][nb]);
                }
            }
        }
        // ---- first-layer weight gradient: dW0[i][k] += sum_s dz0[i][s] x[k][s] ; db0 ------------------------------
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) db0[mb] += dz0[mb];
        lds_wave_sync();
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) *reinterpret_cast<f32x4 *>(&T_d[c * LDT + mb * 16 + 4 * g]) = dz0[mb];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) T_x[c * LDX + kmap<PERM>(kk, g)] = xin[kk];
        lds_wave_sync();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            float bj[NB0];
#pragma unroll
            for (int nb = 0; nb < NB0; ++nb) bj[nb] = T_x[(4 * kk + g) * LDX + nb * 16 + c];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const float ai = T_d[(4 * kk + g) * LDT + mb * 16 + c];
#pragma unroll
                for (int nb = 0; nb < NB0; ++nb) accW0[mb][nb] = mfma4(ai, bj[nb], accW0[mb][nb]);
            }
        }
        // ---- input gradient dX^T = W0^T dz0^T ----------------------------------------------------------------------
        if (d_x) {
            // store offset of column col for sample sx = row_base(sx) + col_part[col]: the column parts are 32-bit loop
            // invariants (12 registers, set up before the tile loop), the row base is one 64-bit value per tile.  (Left to
            // itself the compiler hoisted the complete 64-bit per-column offsets: ~70 registers, spilled in the NH = 2 variants.)
            float *row = d_x + (dx_lm_features ? (s << lm_shift) : s * dx_stride);
#pragma unroll
            for (int fb = 0; fb < NB0; ++fb) {
                if (fb >= nfb) break;
                f32x4 acc = zero4;
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int col = colbase + fb * 16 + c;
                        const float wv = W0t[(nb * 16 + 4 * g + r) * IN_PAD + (col < IN_PAD ? col : IN_PAD - 1)];
                        const float wt = col < IN_PAD ? wv : 0.f;  // (clamped index + select: no branch per load)
                        acc = mfma4(wt, dz0[nb][r], acc);
                    }
                // unconditional stores: a column that is not asked for (or a row behind the live count) goes to this lane's word
                // of the sink instead of being skipped -- a skipped store is a saveexec / branch pair per column (ISA, round 6)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float *dst = (valid && (col_ok & (1u << (fb * 4 + r)))) ? row + col_part[fb * 4 + r] : sink;
                    *dst = acc[r];
                }
            }
        }
    }
#undef NSR_S0
#undef NSR_S1
    // ---- this wave's partial gradient, in blob layout ---------------------------------------------------------------
    int c2 = c, g2 = g;  // (opaque: the ~250 store addresses below must not be computed before the tile loop)
    asm volatile("" : "+v"(c2), "+v"(g2));
    float *P = partials + (uint64_t)blockIdx.x * blob_floats;
    float *pW0 = P, *pb0 = pW0 + W * IN_PAD;
    float *pW1 = pb0 + W, *pb1 = pW1 + (NH == 2 ? W * W : 0);
    float *pWl = (NH == 2) ? pb1 + W : pW1;
    float *pbl = pWl + 16 * W;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB0; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = nb * 16 + c2;
                if (col < IN_PAD) pW0[(mb * 16 + 4 * g2 + r) * IN_PAD + col] = accW0[mb][nb][r];
            }
    if constexpr (NH == 2) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) pW1[(mb * 16 + 4 * g2 + r) * W + nb * 16 + c2] = accW1[mb][nb][r];
    }
    // bias / du sums: reduce the D-layout registers over the 16 sample lanes c2 (lanes differing in bits 0..3)
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v0 = db0[mb][r], v1 = db1[mb][r], v2 = du[mb][r];
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) {
                v0 += __shfl_xor(v0, o, 64);
                if (NH == 2) v1 += __shfl_xor(v1, o, 64);
                v2 += __shfl_xor(v2, o, 64);
            }
            db0[mb][r] = v0; db1[mb][r] = v1; du[mb][r] = v2;
        }
    if (c2 == 0) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pb0[mb * 16 + 4 * g2 + r] = db0[mb][r];
                if (NH == 2) pb1[mb * 16 + 4 * g2 + r] = db1[mb][r];
            }
    }
    // dWl (rows o = 4g + r, columns nb*16 + c2); the per-neuron row-0 sums held in D layout (second-order du, tap tiles) join row 0
    {
        lds_wave_sync();
        if (c2 == 0) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) T_a[mb * 16 + 4 * g2 + r] = du[mb][r];
        }
        lds_wave_sync();
        if (g2 == 0) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) accWl[nb][0] += T_a[nb * 16 + c2];
        }
    }
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) pWl[(4 * g2 + r) * W + nb * 16 + c2] = accWl[nb][r];
    // dbl[o]: lane (g2,c2) holds sums of d_out[.][4kk + g2] over ITS sample column c2; reduce over c2
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        float v = dblv[kk];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (c2 == 0) pbl[4 * kk + g2] = v;
    }
}

// grad[k] (+)= sum over the per-wave partials.  One thread per (parameter, segment of 64 partials): the 14 MB of partials are
// read by ~50 k threads (a single pass of 3.4 k threads took 86 us); segments meet through one fp32 atomic each.
constexpr int VRED_SEG = 64;
__global__ void __launch_bounds__(256)
k_vmlp_reduce(const float *__restrict__ partials, float *__restrict__ grad, uint32_t blob_floats, uint32_t n_blocks)
{
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= blob_floats) return;
    const uint32_t b0 = blockIdx.y * VRED_SEG, b1 = min(n_blocks, b0 + VRED_SEG);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    uint32_t b = b0;
    for (; b + 4 <= b1; b += 4) {
        s0 += partials[(uint64_t)(b + 0) * blob_floats + k];
        s1 += partials[(uint64_t)(b + 1) * blob_floats + k];
        s2 += partials[(uint64_t)(b + 2) * blob_floats + k];
        s3 += partials[(uint64_t)(b + 3) * blob_floats + k];
    }
    for (; b < b1; ++b) s0 += partials[(uint64_t)b * blob_floats + k];
    if (b1 > b0) unsafeAtomicAdd(grad + k, (s0 + s1) + (s2 + s3));
}

int check_vmlp(const NsrVmlpDesc *d, const char *who)
{
    NSR_REQUIRE(d != nullptr, "%s: desc is NULL", who);
    NSR_REQUIRE(d->in_pad % 4 == 0 && d->in_pad >= 4 && d->in_pad <= 48, "%s: in_pad=%u unsupported (4..48, x4)", who,
                d->in_pad);
    NSR_REQUIRE(d->n_in >= 1 && d->n_in <= d->in_pad, "%s: n_in=%u > in_pad=%u", who, d->n_in, d->in_pad);
    NSR_REQUIRE(d->n_out >= 1 && d->n_out <= 16, "%s: n_out=%u unsupported (1..16)", who, d->n_out);
    NSR_REQUIRE(d->n_hidden == 1 || d->n_hidden == 2, "%s: n_hidden=%u unsupported (1, 2)", who, d->n_hidden);
    NSR_REQUIRE(d->activation <= 1, "%s: activation=%u unsupported (0 relu, 1 softplus100)", who, d->activation);
    NSR_REQUIRE(d->in_pad == 24 || d->in_pad == 32 || d->in_pad == 36 || d->in_pad == 40,
                "%s: in_pad=%u has no compiled variant (24, 32, 36, 40)", who, d->in_pad);
    return NSR_OK;
}

uint32_t vmlp_blocks(uint32_t n)
{
    const uint32_t tiles = nsr_div_up(n, 16);
    return tiles < 1024u ? (tiles ? tiles : 1u) : 1024u;  // one wave per SIMD on 256 CUs
}

// ---- parameter blob <-> the nn.Linear tensors of a reference VanillaMLP (models/network_utils.py:95-139) ---------------
// The blob is what the MFMA kernels read; the framework side owns weight / (weight_g, weight_v) / bias tensors (old-style
// torch weight_norm, dim = 0: W[r] = g[r] v[r] / |v[r]|).  One wave per blob row: fold builds the padded blob, unfold turns
// the blob's gradient into the gradients of those tensors (through the weight-norm fold) -- two launches instead of the
// ~45 elementwise / reduction launches the same arithmetic costs through torch autograd.
struct VanillaLayers {
    const float *v[3];     // weight_v (weight-normed layer) or weight
    const float *g[3];     // weight_g or NULL
    const float *bias[3];
    float *grad_v[3], *grad_g[3], *grad_bias[3];
    uint32_t n_out[3], n_in[3];   // logical dims of layer l
    uint32_t rows[3], cols[3];    // padded dims in the blob (64 x in_pad | 64 x 64 | 16 x 64)
    uint32_t w_off[3], b_off[3];  // float offsets of W_l and b_l in the blob
    uint32_t n_layers;
};

__global__ void __launch_bounds__(256)
k_vmlp_fold(const VanillaLayers L, float *__restrict__ blob)
{
    uint32_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    uint32_t l = 0;
    while (l < L.n_layers && row >= L.rows[l]) { row -= L.rows[l]; ++l; }
    if (l >= L.n_layers) return;
    const uint32_t n_in = L.n_in[l], cols = L.cols[l];
    float *w = blob + L.w_off[l] + (uint64_t)row * cols;
    if (row >= L.n_out[l]) {  // padding row of the output layer
        for (uint32_t c = lane; c < cols; c += 64) w[c] = 0.f;
        if (lane == 0) blob[L.b_off[l] + row] = 0.f;
        return;
    }
    const float *v = L.v[l] + (uint64_t)row * n_in;
    float scale = 1.f, inv_norm = 1.f;
    const bool wn = L.g[l] != nullptr;
    if (wn) {
        float ss = 0.f;
        for (uint32_t c = lane; c < n_in; c += 64) ss += v[c] * v[c];
        ss = wave_sum(ss);
        scale = L.g[l][row];
        inv_norm = sqrtf(ss);
    }
    for (uint32_t c = lane; c < cols; c += 64) w[c] = c < n_in ? (wn ? scale * v[c] / inv_norm : v[c]) : 0.f;
    if (lane == 0) blob[L.b_off[l] + row] = L.bias[l][row];
}

__global__ void __launch_bounds__(256)
k_vmlp_unfold(const VanillaLayers L, const float *__restrict__ grad_blob, int accumulate)
{
    uint32_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    uint32_t l = 0;
    while (l < L.n_layers && row >= L.rows[l]) { row -= L.rows[l]; ++l; }
    if (l >= L.n_layers || row >= L.n_out[l]) return;
    const uint32_t n_in = L.n_in[l], cols = L.cols[l];
    const float *dw = grad_blob + L.w_off[l] + (uint64_t)row * cols;
    float *gv = L.grad_v[l] + (uint64_t)row * n_in;
    if (lane == 0) {
        const float db = grad_blob[L.b_off[l] + row];
        L.grad_bias[l][row] = accumulate ? L.grad_bias[l][row] + db : db;
    }
    if (!L.g[l]) {
        for (uint32_t c = lane; c < n_in; c += 64) gv[c] = accumulate ? gv[c] + dw[c] : dw[c];
        return;
    }
    // W = g v / |v|:  dL/dg = dW . v / |v| ;  dL/dv = (g / |v|) (dW - v (dW . v) / |v|^2)
    const float *v = L.v[l] + (uint64_t)row * n_in;
    float ss = 0.f, dot = 0.f;
    for (uint32_t c = lane; c < n_in; c += 64) { ss += v[c] * v[c]; dot += dw[c] * v[c]; }
    ss = wave_sum(ss);
    dot = wave_sum(dot);
    const float norm = sqrtf(ss), g = L.g[l][row];
    if (lane == 0) {
        const float dg = dot / norm;
        L.grad_g[l][row] = accumulate ? L.grad_g[l][row] + dg : dg;
    }
    for (uint32_t c = lane; c < n_in; c += 64) {
        const float d = (g / norm) * (dw[c] - v[c] * (dot / (norm * norm)));
        gv[c] = accumulate ? gv[c] + d : d;
    }
}

}  // namespace

extern "C" uint64_t nsr_vmlp_blob_floats(const NsrVmlpDesc *d)
{
    if (!d) return 0;
    return (uint64_t)W * d->in_pad + W + (d->n_hidden == 2 ? (uint64_t)W * W + W : 0) + 16 * W + 16;
}

extern "C" uint64_t nsr_vmlp_backward_workspace_floats(const NsrVmlpDesc *d, uint32_t n)
{
    return d ? (uint64_t)vmlp_blocks(n) * nsr_vmlp_blob_floats(d) : 0;
}

#define VMLP_DISPATCH(KERNEL, ...)                                                                                      \
    do {                                                                                                                \
        bool done_ = false;                                                                                             \
        VMLP_CASE(KERNEL, 6, __VA_ARGS__) VMLP_CASE(KERNEL, 8, __VA_ARGS__) VMLP_CASE(KERNEL, 9, __VA_ARGS__)           \
        VMLP_CASE(KERNEL, 10, __VA_ARGS__)                                                                              \
        if (!done_) { nsr_set_error("vmlp: no variant"); return NSR_ERR_INVALID; }                                      \
    } while (0)

// the 3 + 32-column SDF input takes the permuted k assignment (kmap): its loader reads whole 16-byte / 4-byte units
static int check_sdf_encoding(const NsrVmlpDesc *desc, const float *x, const nsr_half *enc, uint32_t enc_stride,
                              const char *who)
{
    if (!enc || desc->in_pad != 36) {
        NSR_REQUIRE(!enc || x, "%s: an encoding without positions needs the 32-column, 36-wide padded input", who);
        return NSR_OK;
    }
    NSR_REQUIRE(x ? desc->n_in == 35 : desc->n_in == 32,
                "%s: a 36-wide padded encoded input is [x (3) | 32 features] (n_in = 35) or, with x == NULL, the 32 features "
                "alone (n_in = 32); got n_in=%u", who, desc->n_in);
    if (enc_stride & 0xC0000000u)
        NSR_REQUIRE((enc_stride & 0xffu) == 2u && ((uintptr_t)enc & 3) == 0,
                    "%s: level- / tile-major encodings are read as 2-feature levels (F=%u)", who, enc_stride & 0xffu);
    else
        NSR_REQUIRE(enc_stride % 8 == 0 && ((uintptr_t)enc & 15) == 0,
                    "%s: a row-major encoding needs a 16-byte aligned buffer and a stride that is a multiple of 8 halfs", who);
    return NSR_OK;
}

extern "C" int nsr_vmlp_forward(const NsrVmlpDesc *desc, const float *blob, const float *x, uint32_t x_stride,
                                const nsr_half *enc, uint32_t enc_stride, float *out, float *out_col0, float *g_in,
                                uint32_t n, uint32_t n_full, const int32_t *n_dev, void *stream)
{
    if (int rc = check_vmlp(desc, "nsr_vmlp_forward")) return rc;
    if (int rc = check_sdf_encoding(desc, x, enc, enc_stride, "nsr_vmlp_forward")) return rc;
    if (n == 0) return NSR_OK;
    NSR_REQUIRE(blob && (x || enc) && (out || n_full == 0), "nsr_vmlp_forward: NULL pointer");
    NSR_REQUIRE(n_full <= n && (n_full == n || out_col0), "nsr_vmlp_forward: rows beyond n_full need out_col0");
    NSR_REQUIRE(!g_in || desc->n_hidden == 1, "nsr_vmlp_forward: the input gradient is implemented for one hidden layer");
    const uint32_t blocks = vmlp_blocks(n) * 2 > nsr_div_up(n, 16) ? nsr_div_up(n, 16) : vmlp_blocks(n) * 2;
    const int ks = desc->in_pad / 4, nh = desc->n_hidden, act = desc->activation;
    const bool sdf = enc != nullptr, gin = g_in != nullptr;
    const __half *e = (const __half *)enc;
#define VMLP_CASE(KERNEL, KSV, ...)                                                                                     \
    if (!done_ && ks == KSV) {                                                                                          \
        done_ = true;                                                                                                   \
        if (nh == 1 && act == 0 && !sdf && !gin) hipLaunchKernelGGL((KERNEL<KSV, 1, 0, false, false>), __VA_ARGS__);    \
        else if (nh == 2 && act == 0 && !sdf && !gin) hipLaunchKernelGGL((KERNEL<KSV, 2, 0, false, false>), __VA_ARGS__); \
        else if (nh == 1 && act == 1 && sdf && !gin) hipLaunchKernelGGL((KERNEL<KSV, 1, 1, true, false>), __VA_ARGS__); \
        else if (nh == 1 && act == 0 && sdf && !gin && KSV == 9) hipLaunchKernelGGL((KERNEL<9, 1, 0, true, false>), __VA_ARGS__); \
        else if (nh == 1 && act == 1 && sdf && gin) hipLaunchKernelGGL((KERNEL<KSV, 1, 1, true, true>), __VA_ARGS__);   \
        else if (nh == 1 && act == 1 && !sdf && !gin) hipLaunchKernelGGL((KERNEL<KSV, 1, 1, false, false>), __VA_ARGS__); \
        else if (nh == 1 && act == 1 && !sdf && gin) hipLaunchKernelGGL((KERNEL<KSV, 1, 1, false, true>), __VA_ARGS__); \
        else { nsr_set_error("nsr_vmlp_forward: combination not compiled (n_hidden=%d act=%d sdf=%d grad=%d)", nh, act, \
                             (int)sdf, (int)gin); return NSR_ERR_INVALID; }                                             \
    }
    VMLP_DISPATCH(k_vmlp_forward, dim3(blocks), dim3(64), 0, (hipStream_t)stream, blob, x, x_stride, e, enc_stride,
                  desc->n_in, desc->n_out, out, out_col0, g_in, n, n_full, n_dev);
#undef VMLP_CASE
    NSR_CHECK_LAUNCH("nsr_vmlp_forward");
    return NSR_OK;
}

extern "C" int nsr_vmlp_backward(const NsrVmlpDesc *desc, const float *blob, const float *x, uint32_t x_stride,
                                 const nsr_half *enc, uint32_t enc_stride, const float *d_out, const float *d_out_col0,
                                 const float *p_in, float *d_x, uint32_t dx_stride, uint32_t dx_first, uint32_t dx_count,
                                 uint32_t dx_level_major_features, float *grad_blob, int accumulate, float *partials,
                                 uint32_t n, uint32_t n_full, const int32_t *n_dev, void *stream)
{
    if (int rc = check_vmlp(desc, "nsr_vmlp_backward")) return rc;
    NSR_REQUIRE(n == 0 || x, "nsr_vmlp_backward: x is NULL (the encoding-only input is a forward-only mode)");
    if (int rc = check_sdf_encoding(desc, x, enc, enc_stride, "nsr_vmlp_backward")) return rc;
    NSR_REQUIRE(blob && grad_blob && partials && (n == 0 || x), "nsr_vmlp_backward: NULL pointer");
    NSR_REQUIRE(n_full <= n && (n_full == 0 || d_out) && (n_full == n || d_out_col0),
                "nsr_vmlp_backward: d_out / d_out_col0 do not cover the rows");
    NSR_REQUIRE(!p_in || (desc->n_hidden == 1), "nsr_vmlp_backward: second-order terms need one hidden layer");
    NSR_REQUIRE((dx_level_major_features & (dx_level_major_features - 1u)) == 0u,
                "nsr_vmlp_backward: dx_level_major_features must be a power of two (the hash grid's features per level)");
    NSR_REQUIRE(!dx_level_major_features || (uint64_t)n * (dx_count ? dx_count : desc->n_in) < (1ull << 32),
                "nsr_vmlp_backward: level-major d_x offsets are 32-bit (n x columns must stay below 2^32)");
    const uint32_t bf = (uint32_t)nsr_vmlp_blob_floats(desc);
    const uint32_t blocks = vmlp_blocks(n);
    const int ks = desc->in_pad / 4, nh = desc->n_hidden, act = desc->activation;
    const bool sdf = enc != nullptr, second = p_in != nullptr;
    const __half *e = (const __half *)enc;
    if (dx_count == 0) { dx_first = 0; dx_count = desc->n_in; }
#define VMLP_CASE(KERNEL, KSV, ...)                                                                                     \
    if (!done_ && ks == KSV) {                                                                                          \
        done_ = true;                                                                                                   \
        if (nh == 1 && act == 0 && !sdf && !second) hipLaunchKernelGGL((KERNEL<KSV, 1, 0, false, false>), __VA_ARGS__); \
        else if (nh == 2 && act == 0 && !sdf && !second) hipLaunchKernelGGL((KERNEL<KSV, 2, 0, false, false>), __VA_ARGS__); \
        else if (nh == 1 && act == 1 && sdf && !second) hipLaunchKernelGGL((KERNEL<KSV, 1, 1, true, false>), __VA_ARGS__); \
        else if (nh == 1 && act == 1 && sdf && second) hipLaunchKernelGGL((KERNEL<KSV, 1, 1, true, true>), __VA_ARGS__); \
        else if (nh == 1 && act == 1 && !sdf && !second) hipLaunchKernelGGL((KERNEL<KSV, 1, 1, false, false>), __VA_ARGS__); \
        else if (nh == 1 && act == 1 && !sdf && second) hipLaunchKernelGGL((KERNEL<KSV, 1, 1, false, true>), __VA_ARGS__); \
        else { nsr_set_error("nsr_vmlp_backward: combination not compiled (n_hidden=%d act=%d sdf=%d second=%d)", nh,   \
                             act, (int)sdf, (int)second); return NSR_ERR_INVALID; }                                     \
    }
    VMLP_DISPATCH(k_vmlp_backward, dim3(blocks), dim3(64), 0, (hipStream_t)stream, blob, x, x_stride, e, enc_stride,
                  desc->n_in, d_out, d_out_col0, p_in, d_x, dx_stride, dx_first, dx_count, dx_level_major_features,
                  partials, bf, n, n_full, n_dev);
#undef VMLP_CASE
    NSR_CHECK_LAUNCH("nsr_vmlp_backward");
    if (!accumulate)
        NSR_REQUIRE(hipMemsetAsync(grad_blob, 0, bf * sizeof(float), (hipStream_t)stream) == hipSuccess,
                    "nsr_vmlp_backward: hipMemsetAsync failed");
    hipLaunchKernelGGL(k_vmlp_reduce, dim3(nsr_div_up(bf, 256), nsr_div_up(blocks, VRED_SEG)), dim3(256), 0,
                       (hipStream_t)stream, partials, grad_blob, bf, blocks);
    NSR_CHECK_LAUNCH("nsr_vmlp_reduce");
    return NSR_OK;
}

static int vanilla_layers(const NsrVmlpDesc *d, const NsrVanillaLayer *layers, uint32_t n_layers, bool want_grads,
                          VanillaLayers *L, uint32_t *total_rows)
{
    NSR_REQUIRE(layers && n_layers == d->n_hidden + 1, "vmlp fold: a network with %u hidden layers has %u Linear layers",
                d->n_hidden, d->n_hidden + 1);
    memset(L, 0, sizeof(*L));
    L->n_layers = n_layers;
    uint32_t off = 0, rows = 0;
    for (uint32_t l = 0; l < n_layers; ++l) {
        const bool last = l + 1 == n_layers;
        L->rows[l] = last ? 16 : W;
        L->cols[l] = l == 0 ? d->in_pad : W;
        L->n_out[l] = last ? d->n_out : W;
        L->n_in[l] = l == 0 ? d->n_in : W;
        NSR_REQUIRE(layers[l].n_out == L->n_out[l] && layers[l].n_in == L->n_in[l],
                    "vmlp fold: layer %u is %u x %u, the descriptor says %u x %u", l, layers[l].n_out, layers[l].n_in,
                    L->n_out[l], L->n_in[l]);
        NSR_REQUIRE(layers[l].weight_v && layers[l].bias, "vmlp fold: NULL weight / bias");
        NSR_REQUIRE(!want_grads || (layers[l].grad_v && layers[l].grad_bias && (!layers[l].weight_g || layers[l].grad_g)),
                    "vmlp unfold: NULL gradient tensor");
        L->v[l] = layers[l].weight_v; L->g[l] = layers[l].weight_g; L->bias[l] = layers[l].bias;
        L->grad_v[l] = layers[l].grad_v; L->grad_g[l] = layers[l].grad_g; L->grad_bias[l] = layers[l].grad_bias;
        L->w_off[l] = off;
        off += L->rows[l] * L->cols[l];
        L->b_off[l] = off;
        off += L->rows[l];
        rows += L->rows[l];
    }
    NSR_REQUIRE(off == nsr_vmlp_blob_floats(d), "vmlp fold: blob layout mismatch");
    *total_rows = rows;
    return NSR_OK;
}

extern "C" int nsr_vmlp_fold(const NsrVmlpDesc *desc, const NsrVanillaLayer *layers, uint32_t n_layers, float *blob,
                             void *stream)
{
    if (int rc = check_vmlp(desc, "nsr_vmlp_fold")) return rc;
    NSR_REQUIRE(blob, "nsr_vmlp_fold: NULL blob");
    VanillaLayers L;
    uint32_t rows = 0;
    if (int rc = vanilla_layers(desc, layers, n_layers, false, &L, &rows)) return rc;
    hipLaunchKernelGGL(k_vmlp_fold, dim3(nsr_div_up(rows, 4)), dim3(256), 0, (hipStream_t)stream, L, blob);
    NSR_CHECK_LAUNCH("nsr_vmlp_fold");
    return NSR_OK;
}

extern "C" int nsr_vmlp_unfold_gradient(const NsrVmlpDesc *desc, const NsrVanillaLayer *layers, uint32_t n_layers,
                                        const float *grad_blob, int accumulate, void *stream)
{
    if (int rc = check_vmlp(desc, "nsr_vmlp_unfold_gradient")) return rc;
    NSR_REQUIRE(grad_blob, "nsr_vmlp_unfold_gradient: NULL gradient blob");
    VanillaLayers L;
    uint32_t rows = 0;
    if (int rc = vanilla_layers(desc, layers, n_layers, true, &L, &rows)) return rc;
    hipLaunchKernelGGL(k_vmlp_unfold, dim3(nsr_div_up(rows, 4)), dim3(256), 0, (hipStream_t)stream, L, grad_blob,
                       accumulate);
    NSR_CHECK_LAUNCH("nsr_vmlp_unfold_gradient");
    return NSR_OK;
}
