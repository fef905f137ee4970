// tcnn.NetworkWithInputEncoding (models/network_utils.py:209-214: HashGrid -> FullyFusedMLP with one flat parameter) as ONE
// forward kernel: the wave that runs the register-chained MFMA MLP of csrc/mlp.hip builds its first B operand itself.
//
// In the transposed layout of mlp.hip lane (n = lane & 15, g = lane >> 4) feeds input columns [8g, 8g + 8) of sample n
// of its 16-sample tile into the first v_mfma_f32_16x16x32_f16.  With F features per level those are the 8 / F levels
// [g * 8 / F, (g + 1) * 8 / F): the lane locates its sample on exactly those levels, gathers their 8 corners each (32
// independent 4-byte gathers in flight at F = 2), blends in fp32, rounds to fp16 -- the very value the two-launch path
// stores and re-loads -- and the MFMA chain runs on registers.  The 64 B / sample of encoded features never travel unless
// the caller asks for them (training: the weight gradient of the first layer needs them).
//
// What this gives up is the XCD placement of csrc/hashgrid.hip (XCD b % 8 serves levels {b % 8, b % 8 + 8}: 4 MiB of table
// per 4 MiB L2): here every wave touches all levels, so the fine levels are served by the Infinity Cache instead of the
// L2.  Which wins depends on the launch size; tools/grid_mlp_ab.py measures both and nsr_hip/ops.py picks by n
// (DESIGN.md section 4).
#include "nsr_common.h"
#include "hashgrid_geom.h"
#include "mlp_frag.h"

namespace {

template <int F, int NH>
__global__ void __launch_bounds__(MLP_BLOCK)
k_grid_mlp_forward(const float *__restrict__ x, const __half *__restrict__ table, const __half *__restrict__ W_,
                   __half *__restrict__ out, __half *__restrict__ acts, __half *__restrict__ enc, uint32_t enc_stride,
                   int enc_level_major, uint32_t n, uint32_t mask_count, uint32_t n_in, int out_act, const NsrGridDesc d,
                   const int32_t *__restrict__ n_dev)
{
    constexpr int IN_PAD = 32, LPL = 8 / F;  // levels per lane
    const uint32_t n_live = live_count(n, n_dev);
    const int lane = threadIdx.x & 63, nl = lane & 15, g = lane >> 4;
    const uint32_t wave = (blockIdx.x * MLP_BLOCK + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * MLP_BLOCK) >> 6;
    const uint32_t n_tiles = (n_live + 15) / 16;
    const _Float16 *W = reinterpret_cast<const _Float16 *>(W_);

    // ---- this lane's levels: geometry into registers (selected from the kernel argument by compares, once) ----
    LevelGeom geo[LPL];
    bool live[LPL];
#pragma unroll
    for (int i = 0; i < LPL; ++i) {
        const uint32_t level = (uint32_t)(g * LPL + i);
        geo[i].scale = 0.f; geo[i].res = 1u; geo[i].size = 8u; geo[i].offset = 0u; geo[i].dense = true;
#pragma unroll
        for (uint32_t l = 0; l < NSR_MAX_LEVELS; ++l) {
            if (l == level && l < d.n_levels) geo[i] = load_level(d, l);
        }
        live[i] = level < d.n_levels && level < mask_count;
    }

    // ---- weights -> registers (as k_mlp_forward<2, NH>) ----
    half8 a0[4];
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) a0[ob] = load_a_natural(W, IN_PAD, ob * 16 + nl, 0, g, IN_PAD);
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
        const uint32_t sc = valid ? s : 0u;
        const float x0 = x[3ull * sc], x1 = x[3ull * sc + 1], x2 = x[3ull * sc + 2];
        half8 b;
#pragma unroll
        for (int i = 0; i < LPL; ++i) {
            float v[F];
#pragma unroll
            for (int f = 0; f < F; ++f) v[f] = 0.f;
            if (live[i] && valid) encode_level_from<F>(table + (uint64_t)geo[i].offset * F, geo[i], x0, x1, x2, v);
#pragma unroll
            for (int f = 0; f < F; ++f) {
                const int c = 8 * g + i * F + f;
                // (columns [n_in, in_pad) of the MLP input are the constant 1, as in load_x8)
                b[i * F + f] = c < (int)n_in ? (_Float16)__half2float(__float2half_rn(v[f])) : (_Float16)1;
            }
        }
        if (!valid) {
#pragma unroll
            for (int j = 0; j < 8; ++j) b[j] = (_Float16)0;
        }
        if (enc && valid) {
            if (enc_level_major) {
#pragma unroll
                for (int i = 0; i < LPL; ++i) {
                    const uint32_t level = (uint32_t)(g * LPL + i);
                    if (level >= d.n_levels) continue;
                    _Float16 *dst = reinterpret_cast<_Float16 *>(enc) + ((uint64_t)level * n + s) * F;
#pragma unroll
                    for (int f = 0; f < F; ++f) dst[f] = b[i * F + f];
                }
            } else {
                _Float16 *dst = reinterpret_cast<_Float16 *>(enc) + (uint64_t)s * enc_stride + 8 * g;
                if (8 * g + 8 <= (int)n_in) {
                    *reinterpret_cast<half8 *>(dst) = b;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (8 * g + j < (int)n_in) dst[j] = b[j];
                }
            }
        }
        f32x4 acc[4];
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) acc[ob] = mfma32(a0[ob], b, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int h = 0; h < NH; ++h) {
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


// measured (tools/grid_mlp_ab.py): 512 workgroups beat 2048 / 8192 at every size -- a wave fetches 14 KB of weights before its
// first tile, and more than ~8 waves per CU only add gather requests in flight to an L2 that is already missing
constexpr uint32_t g_max_blocks = 512;

}  // namespace

#define GM_DISPATCH(F_, NH_, ...)                                                  \
    switch ((F_) * 10 + (NH_)) {                                                   \
    case 11: { constexpr int F = 1, NH = 1; __VA_ARGS__; } break;                  \
    case 12: { constexpr int F = 1, NH = 2; __VA_ARGS__; } break;                  \
    case 21: { constexpr int F = 2, NH = 1; __VA_ARGS__; } break;                  \
    case 22: { constexpr int F = 2, NH = 2; __VA_ARGS__; } break;                  \
    case 41: { constexpr int F = 4, NH = 1; __VA_ARGS__; } break;                  \
    case 42: { constexpr int F = 4, NH = 2; __VA_ARGS__; } break;                  \
    case 81: { constexpr int F = 8, NH = 1; __VA_ARGS__; } break;                  \
    default: { constexpr int F = 8, NH = 2; __VA_ARGS__; } break;                  \
    }

extern "C" int nsr_grid_mlp_supported(const NsrGridDesc *grid, const NsrMlpDesc *mlp)
{
    if (!grid || !mlp) return 0;
    const uint32_t F = grid->n_features;
    if (!(F == 1 || F == 2 || F == 4 || F == 8)) return 0;
    if (grid->n_levels < 1 || grid->n_levels > NSR_MAX_LEVELS || grid->n_levels * F > 32) return 0;
    return mlp->in_pad == 32 && mlp->n_in == grid->n_levels * F && mlp->out_pad == 16 && mlp->n_hidden >= 1 &&
           mlp->n_hidden <= 2 && mlp->output_activation <= NSR_ACT_SIGMOID;
}

extern "C" int nsr_grid_mlp_forward(const float *x, const nsr_half *table, const nsr_half *weights, nsr_half *out,
                                    nsr_half *acts, nsr_half *enc, uint32_t enc_stride, int enc_level_major, uint32_t n,
                                    uint32_t level_mask_count, const NsrGridDesc *grid, const NsrMlpDesc *mlp,
                                    const int32_t *n_dev, void *stream)
{
    NSR_REQUIRE(grid && mlp, "nsr_grid_mlp_forward: desc is NULL");
    NSR_REQUIRE(nsr_grid_mlp_supported(grid, mlp),
                "nsr_grid_mlp_forward: needs n_levels * n_features == n_in <= 32 == in_pad, out_pad 16, 1-2 hidden layers "
                "(got L=%u F=%u n_in=%u in_pad=%u hidden=%u): use nsr_hashgrid_forward + nsr_mlp_forward",
                grid->n_levels, grid->n_features, mlp->n_in, mlp->in_pad, mlp->n_hidden);
    if (n == 0) return NSR_OK;
    NSR_REQUIRE(x && table && weights && out, "nsr_grid_mlp_forward: NULL pointer");
    NSR_REQUIRE(!enc || enc_level_major || (enc_stride >= mlp->n_in && enc_stride % 8 == 0 && ((uintptr_t)enc & 15) == 0),
                "nsr_grid_mlp_forward: a row-major encoding output needs a 16-byte aligned buffer and a stride that is a "
                "multiple of 8 halfs");
    const uint32_t n_tiles = (n + 15) / 16;
    uint32_t blocks = (n_tiles + WAVES - 1) / WAVES;
    if (blocks > g_max_blocks) blocks = g_max_blocks;
    GM_DISPATCH(grid->n_features, mlp->n_hidden,
                hipLaunchKernelGGL((k_grid_mlp_forward<F, NH>), dim3(blocks), dim3(MLP_BLOCK), 0, (hipStream_t)stream, x,
                                   (const __half *)table, (const __half *)weights, (__half *)out, (__half *)acts,
                                   (__half *)enc, enc_stride, enc_level_major, n, level_mask_count, mlp->n_in,
                                   (int)mlp->output_activation, *grid, n_dev));
    NSR_CHECK_LAUNCH("nsr_grid_mlp_forward");
    return NSR_OK;
}


// Backward of the pair: nothing to fuse into one kernel -- the MLP's data gradient already leaves k_mlp_dgrad level-major,
// exactly as the owner-computes table backward reads it, so the encoding's gradient makes one trip through HBM and no
// transpose.  One call = dgrad + weight gradients + item binning + owner accumulation.
// workspace (floats): [MLP partials][level-major d_enc: L*F*n][table-backward workspace]
extern "C" uint64_t nsr_grid_mlp_backward_workspace_floats(const NsrGridDesc *grid, const NsrMlpDesc *mlp, uint32_t n)
{
    if (!nsr_grid_mlp_supported(grid, mlp)) return 0;
    return nsr_mlp_backward_workspace_floats(mlp, n) + (uint64_t)grid->n_levels * grid->n_features * n +
           nsr_hashgrid_backward_params_workspace_floats(grid, n) + 8;
}

extern "C" int nsr_grid_mlp_backward(const void *dout, int dout_is_f32, uint32_t dout_stride, const nsr_half *out,
                                     const float *x, const nsr_half *enc, uint32_t enc_stride, int enc_level_major,
                                     const nsr_half *acts, const nsr_half *weights, float *grad_weights, float *grad_table,
                                     float *workspace, uint32_t n, uint32_t level_mask_count, float grad_scale,
                                     const NsrGridDesc *grid, const NsrMlpDesc *mlp, void *stream)
{
    NSR_REQUIRE(grid && mlp && nsr_grid_mlp_supported(grid, mlp), "nsr_grid_mlp_backward: unsupported grid / MLP pair");
    if (n == 0) return NSR_OK;
    NSR_REQUIRE(dout && x && enc && acts && weights && workspace && grad_table, "nsr_grid_mlp_backward: NULL pointer");
    const uint32_t F = grid->n_features;
    float *partials = workspace;
    float *d_enc = partials + ((nsr_mlp_backward_workspace_floats(mlp, n) + 3ull) & ~3ull);
    float *grid_ws = d_enc + (((uint64_t)grid->n_levels * F * n + 3ull) & ~3ull);
    if (int rc = nsr_mlp_backward_ex(dout, dout_is_f32, dout_stride, nullptr, out, enc, 0, enc_stride,
                                     enc_level_major ? F : 0u, acts, weights, grad_weights, d_enc, 0, F, partials, n,
                                     grad_scale, mlp, nullptr, stream))
        return rc;
    return nsr_hashgrid_backward_params_owner(x, d_enc, 2, 0, grad_table, grid_ws, n, level_mask_count, 1.0f, 0, grid,
                                              nullptr, stream);
}
