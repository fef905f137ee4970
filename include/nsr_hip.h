/* nsr_hip.h -- C ABI of libnsr_hip.so: the MI355X (gfx950) kernels behind the `tinycudann` and
 * `nerfacc` Python surfaces that bennyguo/instant-nsr-pl calls.
 *
 * The reference never binds a C symbol: its FFI for this path is the *Python import surface* of two
 * third-party CUDA packages.  Each entry point below names the reference call
 * site (file:line under /root/reference) whose third-party kernel it replaces.  The Python packages in
 * instant-nsr-pl_amd/{tinycudann,nerfacc}/ bind these with ctypes.
 *
 * Conventions
 *   - plain pointers + sizes; every pointer is DEVICE memory owned by the caller unless marked host;
 *   - the library never allocates or frees device memory and keeps no global device state;
 *   - every function enqueues on `stream` (a hipStream_t passed as void*) and returns without syncing;
 *   - return 0 on success, <0 on error; nsr_last_error() gives a thread-local message;
 *   - "accumulate" outputs (grad_table, grad_params) are ADDED to: the caller zeroes them;
 *   - half = IEEE binary16 (uint16_t storage).
 */
#ifndef NSR_HIP_H
#define NSR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NSR_OK 0
#define NSR_ERR_INVALID (-1)
#define NSR_ERR_LAUNCH (-2)
#define NSR_ERR_UNSUPPORTED (-3)

#define NSR_MAX_LEVELS 32

typedef uint16_t nsr_half;

const char *nsr_last_error(void);
/* ABI version of this header (bumped on any signature change) */
int nsr_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * Multiresolution hash grid  -- replaces tcnn.Encoding(HashGrid)
 *   constructed at models/network_utils.py:47,90 and (fused) :209 ; evaluated at
 *   models/geometry.py:124,134,169,195,214
 * ------------------------------------------------------------------------------------------------ */
typedef struct NsrGridDesc {
    uint32_t n_levels;          /* L  (<= NSR_MAX_LEVELS) */
    uint32_t n_features;        /* F  in {1,2,4,8} */
    uint32_t log2_hashmap_size; /* T = 2^log2 */
    uint32_t base_resolution;
    float per_level_scale;
    uint32_t n_entries;                 /* sum of size[] ; n_params = n_entries * F */
    float scale[NSR_MAX_LEVELS];        /* fp32: exp2f(l*log2f(s))*base - 1 */
    uint32_t resolution[NSR_MAX_LEVELS]; /* ceilf(scale)+1 */
    uint32_t size[NSR_MAX_LEVELS];      /* entries in level (dense: res^3 rounded up to 8; capped at T) */
    uint32_t offset[NSR_MAX_LEVELS + 1]; /* entry offset of each level */
} NsrGridDesc;

/* host-only: fill `out` (host memory) with fp32 level geometry computed with log2f/exp2f/ceilf */
int nsr_hashgrid_make_desc(NsrGridDesc *out, uint32_t n_levels, uint32_t n_features,
                           uint32_t log2_hashmap_size, uint32_t base_resolution, float per_level_scale);

/* y[n, y_stride] (half) <- encode(x[n,3] in [0,1], table[n_entries*F] half).  L*F columns written.
 * `level_mask_count`: levels >= this count are written as zeros WITHOUT touching the table
 * (ProgressiveBandHashGrid, models/network_utils.py:55-65); pass n_levels for the plain encoding. */
int nsr_hashgrid_forward(const float *x, const nsr_half *table, nsr_half *y, uint32_t n, uint32_t y_stride,
                         uint32_t level_mask_count, const NsrGridDesc *desc, void *stream);

/* same with a choice of output layout, y_level_major =
 *   0: row-major [n, y_stride] (the tcnn API's);
 *   1: LEVEL-major [L][n][F] halfs (y_stride ignored): every wavefront stores 64*F consecutive halfs instead of 64 scattered
 *      F-half pieces (the fused NeRF step's layout; nsr_mlp_forward_ex reads it);
 *   2: TILE-major [ceil(n/16)][L][16][F] halfs (y_stride ignored; the buffer holds ceil(n/16)*16 rows): the 16 rows of an
 *      MFMA tile are one contiguous block and each level's 16*F halfs are contiguous inside it -- 64-B stores for the
 *      encode, 2-3 runs per load for the fp32 MLP kernels (nsr_vmlp_*: enc_stride = 0x40000000 | F; the fused NeuS steps).
 * nsr_hashgrid_forward_jac and nsr_hashgrid_forward_taps (rows = the 7n points) take the same values. */
int nsr_hashgrid_forward_ex(const float *x, const nsr_half *table, nsr_half *y, uint32_t n, uint32_t y_stride,
                            int y_level_major, uint32_t level_mask_count, const NsrGridDesc *desc,
                            const int32_t *n_dev, void *stream);

/* DEVICE-SIDE ROW COUNTS.  Entry points with an (n, n_dev) pair launch for the CAPACITY n and use n for every array
 * stride; when n_dev (device int32[1]) is not NULL only the first min(*n_dev, n) rows are live.  A whole training step
 * can then be queued without the host ever reading a sample count (csrc/step.hip, nsr/fused.py). */

/* grad_table[n_entries*F] (fp32, accumulate) += scatter(dy).  dy_is_f32: 0 = half, 1 = float.
 * grad_scale multiplies dy on load (pass 1.0f). */
int nsr_hashgrid_backward_params(const float *x, const void *dy, int dy_is_f32, uint32_t dy_stride,
                                 float *grad_table, uint32_t n, uint32_t level_mask_count, float grad_scale,
                                 const NsrGridDesc *desc, void *stream);

/* Same result WITHOUT global atomics ("owner computes", the default path): every workgroup owns a slice of one
 * level's gradient in LDS, scans all samples, and stores its slice once.  dy_layout: 0 = half row-major
 * [n,dy_stride], 1 = float row-major, 2 = float level-major [L][n][F] (dy_stride ignored, no workspace needed).
 * workspace: nsr_hashgrid_backward_params_workspace_floats() floats (always required: chunk slabs of the small
 * dense levels + the level-major copy of a row-major dy).
 * accumulate=0 OVERWRITES grad_table (every entry is written, so the caller need not zero it); 1 adds. */
uint64_t nsr_hashgrid_backward_params_workspace_floats(const NsrGridDesc *desc, uint32_t n);
int nsr_hashgrid_backward_params_owner(const float *x, const void *dy, int dy_layout, uint32_t dy_stride,
                                       float *grad_table, float *workspace, uint32_t n, uint32_t level_mask_count,
                                       float grad_scale, int accumulate, const NsrGridDesc *desc, const int32_t *n_dev,
                                       void *stream);

/* The same in two phases, so that a caller can overlap the first with other work: _bin needs only the positions (it
 * sorts the (sample, corner pair) items by owning slice into `workspace`, one launch), _accumulate needs dy and that workspace. */
int nsr_hashgrid_backward_params_owner_bin(const float *x, float *workspace, uint32_t n, uint32_t level_mask_count,
                                           const NsrGridDesc *desc, const int32_t *n_dev, void *stream);
/* ... when the items will be consumed by nsr_hashgrid_backward_params_owner_with_second_order* with binned != 0: the slice
 * configuration of a launch depends on the point count AND on the kind of pass (second-order items cost about twice a plain
 * one's), and binning and accumulation have to agree on it. */
int nsr_hashgrid_backward_params_owner_bin_second_order(const float *x, float *workspace, uint32_t n,
                                                        uint32_t level_mask_count, const NsrGridDesc *desc,
                                                        const int32_t *n_dev, void *stream);
int nsr_hashgrid_backward_params_owner_accumulate(const float *x, const void *dy, int dy_layout, uint32_t dy_stride,
                                                  float *grad_table, float *workspace, uint32_t n,
                                                  uint32_t level_mask_count, float grad_scale, int accumulate,
                                                  const NsrGridDesc *desc, const int32_t *n_dev, void *stream);

/* The owner-computes backward is compiled in two configurations -- 2^11-entry slices x 256 threads (the NeRF step's ~1e5
 * samples per launch) and 2^13 x 1024 (~1e6-point launches of the NeuS steps) -- and a launch picks by its point count:
 * more than `n_points` points -> the large one (default 400,000; 0: always, UINT32_MAX: never).  Returns the previous
 * threshold.  Same results either way (bit-identical on the levels that are not split into chunk slabs). */
uint32_t nsr_hashgrid_owner_large_from(uint32_t n_points);

/* Run-time knobs of the owner-computes decomposition (A/B switches; the gradient is the same bits under every setting).
 * key 0: placement of the (level, slice, chunk) work units on the eight XCDs -- 0: dealt round-robin; 1: XCD k
 * takes the k-th contiguous, cost-balanced range of the unit list (one or two levels per L2: the least x / dy fetch); 2: level
 * l belongs to the XCD pair l mod 4, a pair's units are dealt to its two XCDs (a level's dy is fetched by 2 L2s instead of 8);
 * 3: the same per-pair lists, finest level first, but a workgroup claims its unit when it starts (an atomic cursor per pair) and
 * takes from the next pair's list once its own is empty; 4 (default): 3 in the 2^13 x 1024 configuration, 2 in the 2^11 x 256 one.
 * key 1 / 2: weight of a unit's write-out share / item share in the cost balance of placement 1 (default 1 / 3).
 * key 3: log2 of the entries per slice of a dense level (default 11).  key 4: workgroups a dense level is cut into at least,
 * slices x item chunks (default 64).  key 5: fp32 merge of runs of same-entry lanes before the LDS atomics -- 0: off, 1: on the
 * chunked dense levels (default; their result is summed from fp32 chunk slabs anyway), 2: on every dense level.
 * Returns the previous value. */
float nsr_hashgrid_owner_tune(int key, float value);
/* _accumulate over the run of levels [level_begin, level_end) only (items binned beforehand, dy level-major fp32
 * [L][n][F]), the gradient written either as fp32 into grad_table or as bf16 (round to nearest even) into grad_bf16 --
 * exactly one of the two is non-NULL; both are indexed like the table (entry 0 of level 0 first) and OVERWRITTEN.
 * bf16 is the transport format of the multi-GPU gradient exchange (nsr/parallel.py): a ray-sharded step
 * launches the finest levels first and starts their reduce-scatter while the coarse levels are still accumulating. */
int nsr_hashgrid_backward_params_owner_accumulate_range(const float *x, const float *dy_level_major, float *grad_table,
                                                        void *grad_bf16, float *workspace, uint32_t n,
                                                        uint32_t level_mask_count, float grad_scale, uint32_t level_begin,
                                                        uint32_t level_end, const NsrGridDesc *desc, const int32_t *n_dev,
                                                        void *stream);

/* dx[n,3] (fp32) = (d y / d x)^T dy  -- the NeuS analytic normal, models/geometry.py:177-180 */
int nsr_hashgrid_backward_input(const float *x, const nsr_half *table, const void *dy, int dy_is_f32,
                                uint32_t dy_stride, float *dx, uint32_t n, uint32_t level_mask_count,
                                const NsrGridDesc *desc, void *stream);

/* double backward of backward_input (eikonal loss through create_graph=True normals,
 * models/geometry.py:177-180 -> systems/neus.py:106).  Given g = dL/d(dx) [n,3]:
 *   d_dy[n, L*F] (fp32, may be NULL)       = J g
 *   grad_table (fp32, accumulate, may be NULL) += d(dx.g)/d table
 *   dx2[n,3] (fp32, may be NULL)           = d(dx.g)/d x                                           */
int nsr_hashgrid_backward_backward_input(const float *x, const nsr_half *table, const void *dy, int dy_is_f32,
                                         uint32_t dy_stride, const float *g, float *d_dy, uint32_t d_dy_stride,
                                         float *grad_table, float *dx2, uint32_t n, uint32_t level_mask_count,
                                         const NsrGridDesc *desc, void *stream);
/* ... with the table gradient through the binned owner-computes path (no global float atomics); workspace of
 * nsr_hashgrid_backward_params_workspace_floats(desc, n) floats; NULL workspace = the atomic kernel above */
int nsr_hashgrid_backward_backward_input_ws(const float *x, const nsr_half *table, const void *dy, int dy_is_f32,
                                            uint32_t dy_stride, const float *g, float *d_dy, uint32_t d_dy_stride,
                                            float *grad_table, float *dx2, float *workspace, uint32_t n,
                                            uint32_t level_mask_count, const NsrGridDesc *desc, void *stream);

/* A/B switch of the forward's decomposition (process-wide, for measurements; default (0, 2)): lds_levels = leading small
 * dense levels (<= 16384 entries) encoded from an LDS copy of their table by persistent workgroups; levels_per_lane = 2:
 * one lane encodes both levels {l, l + 8} its XCD owns.  Results are identical bit for bit. */
int nsr_hashgrid_forward_variant(int lds_levels, int levels_per_lane);

/* Forward that also stores the Jacobian d y / d x per level (jac: level-major fp32 [L][n][F][3], zero on masked levels) from
 * the corner values it already holds, and the two dense products the analytic normal needs from it
 * (models/geometry.py:176-180): dx = J^T dy (first-order input gradient) and d_dy = J g (its double backward) --
 * instead of gathering the table a second and a third time (nsr_hashgrid_backward_input / _backward_backward_input). */
int nsr_hashgrid_forward_jac(const float *x, const nsr_half *table, nsr_half *y, uint32_t n, uint32_t y_stride,
                             int y_level_major, uint32_t level_mask_count, const NsrGridDesc *desc, float *jac,
                             const int32_t *n_dev, void *stream);
int nsr_hashgrid_jac_apply(const float *jac, uint32_t n, const NsrGridDesc *desc, const float *dy, uint32_t dy_stride,
                           float *dx, const float *g, float *d_dy, uint32_t d_dy_stride, const int32_t *n_dev,
                           void *stream);
/* ... optionally leaving a level-major copy [L][n][F] of dy behind (with dx): the layout the table backward's second-order
 * term reads (nsr_hashgrid_backward_params_owner_with_second_order*: dy_stride == 0 means "dy is level-major") */
int nsr_hashgrid_jac_apply_ex(const float *jac, uint32_t n, const NsrGridDesc *desc, const float *dy, uint32_t dy_stride,
                              float *dx, const float *g, float *d_dy, uint32_t d_dy_stride, float *dy_level_major_out,
                              const int32_t *n_dev, void *stream);

/* Encode a sample and its six finite-difference taps in one launch (models/geometry.py:181-197): x7 [7][n][3] as written by
 * nsr_neus_points (row 0 the sample, rows 1 + 2k / 2 + 2k the +-eps taps along axis k: only that axis may differ from the
 * sample); y [7 n][y_stride] half (or level-major [L][7 n][F] with y_level_major), same values as nsr_hashgrid_forward on
 * the 7 n points.  The sample's 8 corners per level are gathered once and shared by the taps that stay in its cell or
 * cross one face of it. */
int nsr_hashgrid_forward_taps(const float *x7, const nsr_half *table, nsr_half *y, uint32_t n, uint32_t y_stride,
                              int y_level_major, uint32_t level_mask_count, const NsrGridDesc *desc, const int32_t *n_dev,
                              void *stream);
/* The same, and the per-(level, sample) crossing masks of the six taps (bit t: tap t + 1 left the sample's cell) are left at
 * the head of tap_workspace (nsr_hashgrid_backward_params_taps_workspace_floats(desc, n) floats) for
 * nsr_hashgrid_backward_params_owner_bin_taps_masked: the stencil mode of the table backward then needs no pass of its own
 * over the 7 n positions x L levels. */
int nsr_hashgrid_forward_taps_masks(const float *x7, const nsr_half *table, nsr_half *y, uint32_t n, uint32_t y_stride,
                                    int y_level_major, uint32_t level_mask_count, const NsrGridDesc *desc,
                                    float *tap_workspace, void *stream);

/* One pass for both table gradients of a NeuS step with analytic normals: grad_table (+)= scatter(dy_first) (first order,
 * dy_first_lm level-major fp32 [L][n][F]) + d(dx.g)/d table (second order; dy row-major fp32 = d sdf / d encoding) */
int nsr_hashgrid_backward_params_owner_with_second_order(const float *x, const float *dy_first_lm, const float *dy,
                                                         uint32_t dy_stride, const float *g, float *grad_table,
                                                         float *workspace, uint32_t n, uint32_t level_mask_count,
                                                         int accumulate, int binned, const NsrGridDesc *desc, void *stream);
/* binned != 0: nsr_hashgrid_backward_params_owner_bin already ran on these positions into `workspace` */

/* ------------------------------------------------------------------------------------------------
 * Spherical harmonics degree 4 -- replaces tcnn.Encoding(SphericalHarmonics), models/texture.py:25
 *   u[n,3] in [0,1] (the reference pre-maps (d+1)/2) -> y[n, y_stride] half, 16 columns
 * ------------------------------------------------------------------------------------------------ */
int nsr_sh4_forward(const float *u, nsr_half *y, uint32_t n, uint32_t y_stride, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Fully fused 64-wide MLP -- replaces tcnn.Network(FullyFusedMLP), models/network_utils.py:181,209
 *   weights: half, row-major [out,in] matrices concatenated: W0[64,in_pad] | (n_hidden-1) x [64,64] |
 *   Wlast[out_pad,64]  (layout documented at models/network_utils.py:142-173); no biases; ReLU.
 *   in_pad in {16,32,48,64}; out_pad == 16; 1 <= n_hidden <= 4.
 * ------------------------------------------------------------------------------------------------ */
#define NSR_ACT_NONE 0
#define NSR_ACT_SIGMOID 1

typedef struct NsrMlpDesc {
    uint32_t n_in;     /* logical input columns (<= in_pad); columns [n_in,in_pad) are the constant 1.0 */
    uint32_t in_pad;
    uint32_t n_out;    /* logical outputs (<= out_pad) */
    uint32_t out_pad;
    uint32_t n_hidden; /* hidden layers of width 64 */
    uint32_t output_activation; /* NSR_ACT_* */
} NsrMlpDesc;

/* x: [n, x_stride] half (x_is_f32=0) or float (x_is_f32=1); out: [n, out_pad] half.
 * acts (may be NULL for inference): n_hidden * [n,64] half post-ReLU activations saved for backward. */
int nsr_mlp_forward(const void *x, int x_is_f32, uint32_t x_stride, const nsr_half *weights, nsr_half *out,
                    nsr_half *acts, uint32_t n, const NsrMlpDesc *desc, void *stream);

/* x_level_major_features = F > 0: x is the level-major fp16 encoding [n_in/F][n][F] (x_is_f32 must be 0) */
int nsr_mlp_forward_ex(const void *x, int x_is_f32, uint32_t x_stride, uint32_t x_level_major_features,
                       const nsr_half *weights, nsr_half *out, nsr_half *acts, uint32_t n, const NsrMlpDesc *desc,
                       const int32_t *n_dev, void *stream);

/* dout: [n, dout_stride] half/float grads w.r.t. the (activated) outputs; out: forward outputs (needed
 * for the sigmoid derivative, may be NULL for NSR_ACT_NONE); x/acts as given to / saved by forward.
 * grad_weights (fp32, accumulate, same layout as weights, may be NULL).
 * dx (may be NULL): [n, dx_stride] float, first n_in columns written.
 * partials: caller-provided fp32 workspace of nsr_mlp_backward_workspace_floats() floats.
 * grad_scale: dout is multiplied by it before the fp16 backward chain and results divided by it. */
uint64_t nsr_mlp_backward_workspace_floats(const NsrMlpDesc *desc, uint32_t n);
int nsr_mlp_backward(const void *dout, int dout_is_f32, uint32_t dout_stride, const nsr_half *out,
                     const void *x, int x_is_f32, uint32_t x_stride, const nsr_half *acts,
                     const nsr_half *weights, float *grad_weights, float *dx, uint32_t dx_stride,
                     float *partials, uint32_t n, float grad_scale, const NsrMlpDesc *desc, void *stream);

/* Extended form used by the fused training step: `dout_extra_col0` (float[n], may be NULL) is added to column 0 of
 * dout on load (the density gradient joining the feature gradient, models/geometry.py:125); with
 * dx_level_major_features = F > 0, dx is written level-major [n_in/F][n][F] (what the owner-computes hash-grid
 * backward reads) and dx_stride is ignored. */
int nsr_mlp_backward_ex(const void *dout, int dout_is_f32, uint32_t dout_stride, const float *dout_extra_col0,
                        const nsr_half *out, const void *x, int x_is_f32, uint32_t x_stride,
                        uint32_t x_level_major_features, const nsr_half *acts,
                        const nsr_half *weights, float *grad_weights, float *dx, uint32_t dx_stride,
                        uint32_t dx_level_major_features, float *partials, uint32_t n, float grad_scale,
                        const NsrMlpDesc *desc, const int32_t *n_dev, void *stream);
/* nsr_mlp_backward_ex in two halves: phases 1 = the dgrad kernel (it also saves what the
 * weight-gradient kernels need in `partials`), 2 = the weight-gradient kernels + their reduction on `stream` (dout may be
 * NULL), 3 = both on `stream`.  The fused NeRF step forks its helper stream once, behind the second network's dgrad. */
int nsr_mlp_backward_phases(const void *dout, int dout_is_f32, uint32_t dout_stride, const float *dout_extra_col0,
                            const nsr_half *out, const void *x, int x_is_f32, uint32_t x_stride,
                            uint32_t x_level_major_features, const nsr_half *acts, const nsr_half *weights,
                            float *grad_weights, float *dx, uint32_t dx_stride, uint32_t dx_level_major_features,
                            float *partials, uint32_t n, float grad_scale, const NsrMlpDesc *desc, const int32_t *n_dev,
                            void *stream, int phases);

/* Round 5 -- the data gradients of BOTH networks of the NeRF step in one kernel: the colour network's input is
 * [16 geometry features | SH4(dir)] (models/texture.py:24-26), so its input gradient's first 16 columns ARE the density
 * network's output gradient (+ d_logit on column 0, models/geometry.py:127); the 16-wide d_feature row stays in the lanes'
 * registers.  d_rgb [n,3], d_logit [n] in; d_enc level-major fp32 [16][n][2] out; the pre-activation gradients are saved in the
 * two networks' backward workspaces where nsr_mlp_backward_phases(..., 1) puts them, so nsr_mlp_backward_phases(..., 2)
 * follows unchanged; every value is bit-identical to the two-launch sequence.  _supported: colour 32 -> 64 x (1..2) -> 3
 * sigmoid, density 32 -> 64 x (1..2) -> 16 linear. */
/* at most `blocks` workgroups per weight-gradient launch (1 .. 512, default 512 or NSR_WGRAD_MAX_BLOCKS; 0 queries); returns the
 * previous cap.  Both halves of one backward must see the same value. */
uint32_t nsr_mlp_wgrad_max_blocks(uint32_t blocks);
int nsr_mlp_dgrad_pair_supported(const NsrMlpDesc *color, const NsrMlpDesc *density);
int nsr_mlp_dgrad_pair(const float *d_rgb, const float *d_logit, const nsr_half *out_color, const nsr_half *acts_color,
                       const nsr_half *w_color, float *partials_color, const nsr_half *acts_density,
                       const nsr_half *w_density, float *partials_density, float *d_enc_level_major, uint32_t n,
                       float grad_scale, const NsrMlpDesc *color, const NsrMlpDesc *density, const int32_t *n_dev,
                       void *stream);


/* ------------------------------------------------------------------------------------------------
 * tcnn.NetworkWithInputEncoding -- models/network_utils.py:209-214 (HashGrid -> FullyFusedMLP, one flat parameter
 * [network | grid]); SURVEY.md section 8(b) `nsr_grid_mlp_forward/backward`.
 *
 * Forward in ONE kernel: the wave that runs the MLP encodes its 16-sample tile into the first MFMA operand itself (lane
 * (sample, g) = levels [g * 8 / F, (g + 1) * 8 / F)); results are bit-identical to nsr_hashgrid_forward + nsr_mlp_forward.
 * Needs n_levels * n_features == mlp.n_in <= 32 == mlp.in_pad, out_pad 16, 1-2 hidden layers (nsr_grid_mlp_supported).
 *   out  [n,16] half; acts (NULL for inference) n_hidden * [n,64] half; enc (NULL, or the encoded features for the
 *   backward: row-major [n, enc_stride] half with enc_stride % 8 == 0, or level-major [L][n][F] with enc_level_major).
 * It trades the XCD placement of the stand-alone encode for one launch and 128 B / sample less traffic: faster for small
 * launches, slower for large ones (DESIGN.md section 4 has the measured crossover; nsr_hip/ops.py picks by n).
 * ------------------------------------------------------------------------------------------------ */
int nsr_grid_mlp_supported(const NsrGridDesc *grid, const NsrMlpDesc *mlp);
int nsr_grid_mlp_forward(const float *x, const nsr_half *table, const nsr_half *weights, nsr_half *out, nsr_half *acts,
                         nsr_half *enc, uint32_t enc_stride, int enc_level_major, uint32_t n, uint32_t level_mask_count,
                         const NsrGridDesc *grid, const NsrMlpDesc *mlp, const int32_t *n_dev, void *stream);
/* Backward of the pair in one call: MLP data gradient written level-major (what the table backward reads: no transpose,
 * one trip through HBM), weight gradients (grad_weights fp32, ACCUMULATED, may be NULL), item binning + owner-computes
 * accumulation into grad_table (fp32, OVERWRITTEN).  dout / out / acts / grad_scale as nsr_mlp_backward; enc as written
 * by the forward.  workspace: nsr_grid_mlp_backward_workspace_floats() floats. */
uint64_t nsr_grid_mlp_backward_workspace_floats(const NsrGridDesc *grid, const NsrMlpDesc *mlp, uint32_t n);
int nsr_grid_mlp_backward(const void *dout, int dout_is_f32, uint32_t dout_stride, const nsr_half *out, const float *x,
                          const nsr_half *enc, uint32_t enc_stride, int enc_level_major, const nsr_half *acts,
                          const nsr_half *weights, float *grad_weights, float *grad_table, float *workspace, uint32_t n,
                          uint32_t level_mask_count, float grad_scale, const NsrGridDesc *grid, const NsrMlpDesc *mlp,
                          void *stream);

/* ------------------------------------------------------------------------------------------------
 * nerfacc 0.3.3 kernels
 * ------------------------------------------------------------------------------------------------ */
#define NSR_CONTRACT_AABB 0
#define NSR_CONTRACT_UN_BOUNDED_TANH 1
#define NSR_CONTRACT_UN_BOUNDED_SPHERE 2

/* nerfacc.intersection.ray_aabb_intersect -- models/neus.py:153 ; inside ray_marching when scene_aabb
 * is given (models/nerf.py:85, models/neus.py:212).  miss => t_min=t_max=1e10 ; t_min clamped >= 0 */
int nsr_ray_aabb_intersect(const float *rays_o, const float *rays_d, const float *aabb /*device[6]*/,
                           float *t_min, float *t_max, uint32_t n_rays, void *stream);

/* nerfacc.ray_marching -- models/nerf.py:83 ; models/neus.py:159,210.   Two-call protocol:
 *   count: num_steps[n_rays] (int32)   <- samples each ray will emit
 *   (caller: exclusive scan -> packed_info[n_rays,2] = (start,count); reads the total back)
 *   write: ray_indices[total] int64, t_starts[total], t_ends[total]
 * roi: device[6]; grid_binary: uint8/bool [res_x*res_y*res_z]; bit-exact fp32 stepping. */
int nsr_ray_march_count(const float *rays_o, const float *rays_d, const float *t_min, const float *t_max,
                        const float *roi, const uint8_t *grid_binary, int res_x, int res_y, int res_z,
                        int contraction, float step_size, float cone_angle, int32_t *num_steps,
                        uint32_t n_rays, void *stream);
int nsr_ray_march_write(const float *rays_o, const float *rays_d, const float *t_min, const float *t_max,
                        const float *roi, const uint8_t *grid_binary, int res_x, int res_y, int res_z,
                        int contraction, float step_size, float cone_angle, const int32_t *packed_info,
                        int64_t *ray_indices, float *t_starts, float *t_ends, uint32_t n_rays, void *stream);

/* Brick-packed variant (the default path of the Python packages): same samples, bit for bit.
 *   nsr_grid_pack_bricks: bool grid [rx,ry,rz] (multiples of 4) -> 4x4x4 bricks of 64 bits + one any-bit per brick
 *     (`bricks` holds nsr_grid_bricks_words64() uint64 words; re-pack whenever the grid changes).
 *   count: with scratch (n_rays * capacity float2, capacity = nsr_ray_march_capacity(roi, step) > 0, AABB type
 *     only) it is the ONLY marching pass and also stores (t0,t1) per ray; write then just packs the rows.
 *     Without scratch (capacity 0) it counts, and write marches a second time (any contraction type). */
uint64_t nsr_grid_bricks_words64(int res_x, int res_y, int res_z);
int nsr_grid_pack_bricks(const uint8_t *grid_binary, int res_x, int res_y, int res_z, uint64_t *bricks, void *stream);
uint32_t nsr_ray_march_capacity(const float *roi_host /*host[6]*/, float step_size);
/* rays carried per 64-lane wave by the brick marcher (1..64; 0 = chosen by the ray count: a wave is as slow as its slowest
 * ray and pays both sides of every divergent branch, and few rays leave most SIMDs idle).  Returns the previous setting. */
uint32_t nsr_ray_march_rays_per_wave(uint32_t rays_per_wave);
/* the wave-per-ray marcher (one ray per wave, 64 candidate samples tested at once; AABB contraction, cone_angle 0; same bits
 * as the lane-per-ray kernel): 0 never, 1 for launches of at most 32,768 rays (default), 2 always.  Returns the previous mode. */
int nsr_ray_march_wave_mode(int mode);
int nsr_ray_march_bricks_count(const float *rays_o, const float *rays_d, const float *t_min, const float *t_max,
                               const float *roi, const uint64_t *bricks, int res_x, int res_y, int res_z,
                               int contraction, float step_size, float cone_angle, int32_t *num_steps, float *scratch,
                               uint32_t capacity, uint32_t n_rays, void *stream);
int nsr_ray_march_bricks_write(const float *rays_o, const float *rays_d, const float *t_min, const float *t_max,
                               const float *roi, const uint64_t *bricks, int res_x, int res_y, int res_z,
                               int contraction, float step_size, float cone_angle, const int32_t *packed_info,
                               const float *scratch, uint32_t capacity, int64_t *ray_indices, float *t_starts,
                               float *t_ends, uint32_t n_rays, void *stream);

/* exclusive scan of num_steps -> packed_info[n_rays,2]; *total (device int32[1]) <- sum */
int nsr_pack_from_counts(const int32_t *num_steps, int32_t *packed_info, int32_t *total, uint32_t n_rays,
                         void *stream);
/* ... for fixed-size sample buffers: with capacity > 0 the packed ranges are truncated to `capacity` samples in total
 * (*total <- min(sum, capacity)); stats (device int32[6], 8-byte aligned, may be NULL): [0] last unclamped sum,
 * [1] largest sum so far (atomic max), [2] number of truncated launches, [4..5] uint64 running sum of the sums */
int nsr_pack_from_counts_capped(const int32_t *num_steps, int32_t *packed_info, int32_t *total, uint32_t n_rays,
                                uint32_t capacity, int32_t *stats, const int32_t *n_active, void *stream);
/* n_active (device int32[1], may be NULL): rays >= *n_active count as empty -- the dead slots of a dynamic ray batch whose
 * marching pass ran before the batch size was known */
/* packed_info[n_rays,2] from sorted ray_indices[n] (binary search per ray) */
int nsr_pack_info(const int64_t *ray_indices, int32_t *packed_info, uint32_t n, uint32_t n_rays, void *stream);

/* contraction / inverse / grid query -- OccupancyGrid._update behind models/nerf.py:55, neus.py:109-111 */
int nsr_contract(const float *x, const float *roi, int contraction, float *out, uint32_t n, void *stream);
int nsr_contract_inv(const float *x, const float *roi, int contraction, float *out, uint32_t n, void *stream);
int nsr_grid_query_u8(const float *x, const float *roi, const uint8_t *grid, int res_x, int res_y, int res_z,
                      int contraction, uint8_t *out, uint32_t n, void *stream);

/* sample positions p = o[ray] + d[ray] * (t0+t1)/2 and per-sample dirs (the gather at
 * models/nerf.py:66-69,95-99 ; models/neus.py:222-227).  dirs_out may be NULL. */
int nsr_sample_positions(const float *rays_o, const float *rays_d, const int64_t *ray_indices,
                         const float *t_starts, const float *t_ends, float *positions, float *dirs_out,
                         uint32_t n, void *stream);

/* nerfacc.render_weight_from_density / render_weight_from_alpha / render_visibility
 *   models/nerf.py:105 ; models/neus.py:181,237 ; inside ray_marching (sigma_fn pruning, nerf.py:87)
 * Segments = contiguous runs of equal ray index; packed_info[n_rays,2] = (start,count).
 *   from_sigma: T_i = exp(-sum_{j<i} sigma_j*(t1_j-t0_j)) ; from_alpha: T_i = prod_{j<i}(1-alpha_j)
 *   backward:   d/d(sigma*dt)_j = -sum_{i>j} gT_i T_i ; d/d alpha_j = that / max(1-alpha_j,1e-10) */
int nsr_transmittance_from_sigma_forward(const int32_t *packed_info, const float *t_starts, const float *t_ends,
                                         const float *sigmas, float *trans, uint32_t n_rays, void *stream);
int nsr_transmittance_from_sigma_backward(const int32_t *packed_info, const float *t_starts, const float *t_ends,
                                          const float *trans, const float *grad_trans, float *grad_sigmas,
                                          uint32_t n_rays, void *stream);
int nsr_transmittance_from_alpha_forward(const int32_t *packed_info, const float *alphas, float *trans,
                                         uint32_t n_rays, void *stream);
int nsr_transmittance_from_alpha_backward(const int32_t *packed_info, const float *alphas, const float *trans,
                                          const float *grad_trans, float *grad_alphas, uint32_t n_rays,
                                          void *stream);

/* nerfacc.accumulate_along_rays -- models/nerf.py:106-108 ; models/neus.py:182-184,238-242
 *   out[r, d] = sum_{i in ray r} w_i * v_i[d]   (values==NULL: D=1, v=1).  Deterministic per-ray reduce. */
int nsr_accumulate_along_rays_forward(const int32_t *packed_info, const float *weights, const float *values,
                                      uint32_t dim, float *out, uint32_t n_rays, void *stream);
int nsr_accumulate_along_rays_backward(const int64_t *ray_indices, const float *weights, const float *values,
                                       uint32_t dim, const float *grad_out, float *grad_weights,
                                       float *grad_values, uint32_t n, void *stream);

/* stream compaction by a byte mask (the boolean filter after render_visibility): writes the kept
 * samples in order; *n_kept (device int32[1]).  Outputs must hold n elements. */
int nsr_compact_samples(const uint8_t *mask, const int64_t *ray_indices, const float *t_starts,
                        const float *t_ends, int64_t *ray_indices_out, float *t_starts_out, float *t_ends_out,
                        int32_t *n_kept, int32_t *block_scratch, uint32_t n, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Reference-owned glue, fused (rows a8-a10, a18 of SURVEY.md section 8a)
 * ------------------------------------------------------------------------------------------------ */
/* contract_to_unisphere, models/geometry.py:17-29 (AABB and UN_BOUNDED_SPHERE) */
int nsr_contract_to_unisphere(const float *x, float radius, int contraction, float *out, uint32_t n,
                              void *stream);
/* trunc_exp(out[:,0] + bias) forward (models/utils.py:53-68 ; models/geometry.py:125-127):
 * mlp_out [n, stride] half -> density[n] float ; feature[n, n_feat] float (may be NULL) */
int nsr_density_activation_forward(const nsr_half *mlp_out, uint32_t stride, uint32_t n_feat, float bias,
                                   float *density, float *feature, uint32_t n, void *stream);
/* NeuS SDF->alpha, models/neus.py:117-139.  inv_s: device float[1] (already exp(10*variance)). */
int nsr_neus_alpha_forward(const float *sdf, const float *normal, const float *dirs, const float *dists,
                           const float *inv_s, float cos_anneal_ratio, float *alpha, uint32_t n, void *stream);
int nsr_neus_alpha_backward(const float *sdf, const float *normal, const float *dirs, const float *dists,
                            const float *inv_s, float cos_anneal_ratio, const float *grad_alpha,
                            float *grad_sdf, float *grad_normal, float *grad_inv_s, /* grad_inv_s: device[1], accumulated */
                            uint32_t n, void *stream);

/* Device-side occupancy refresh (nerfacc 0.3.3 OccupancyGrid._update, reached from models/nerf.py:45-55 every 16th step)
 * without a host sync; csrc/occupancy.hip.  (1) select_cells: all cells (all_cells != 0, capacity >= res^3) or n_uniform
 * cells floor(u_cell * res^3) + the occupied cells of the 4^3-brick bitfield (n_uniform of them, picked with replacement by
 * u_pick, when more than n_uniform are occupied); writes cells[], their jittered positions x_unit[,3] =
 * (cell coordinate + jitter) / res in grid-unit space and the device count n_cells.  all_cells | 2: the grid lives on a
 * sphere-contracted space -- samples with |x_unit - 0.5| >= 0.5 are dropped here (nerfacc grid.py drops them before it
 * evaluates them), n_cells counts the survivors, appended workgroup by workgroup (4,096 slots each, in slot order inside).  jitter: capacity x 3 uniforms;
 * brick_offset: one word per brick; occupied_cells: res^3 words.  (2) the caller evaluates the density network on those
 * positions with n_dev = n_cells.  (3) update: occs_new = occs_old, occs_new[cell] = max(occs_old[cell] * ema_decay,
 * exp(mlp_out[i, 0] + density_bias) * step_size); threshold[0] = min(mean(occs_new), occ_thre); binary = occs_new > it.
 * threshold: 8-byte aligned workspace of 2 + 2*256 floats ([0] the value, the rest partial sums). */
int nsr_occupancy_select_cells(const uint64_t *bricks, int res_x, int res_y, int res_z, const float *u_cell,
                               const float *u_pick, const float *jitter, uint32_t n_uniform, int all_cells,
                               uint32_t capacity, uint32_t *brick_offset, uint32_t *occupied_cells, int32_t *n_occupied,
                               uint32_t *cells, float *x_unit, int32_t *n_cells, void *stream);
int nsr_occupancy_update(const nsr_half *mlp_out, uint32_t stride, float density_bias, float step_size, float ema_decay,
                         float occ_thre, const uint32_t *cells, const float *occs_old, float *occs_new, uint8_t *binary,
                         float *threshold, uint32_t n_total_cells, uint32_t capacity, const int32_t *n_cells, void *stream);

/* occupancy statistic of a density field on a grid in UN_BOUNDED_SPHERE-contracted space (the NeRF++ background grid,
 * models/neus.py:103-106): occ[i] = exp(logit[i] + density_bias) * step_size for samples whose grid-unit position lies
 * inside the unit sphere (|x_unit - 0.5| < 0.5), -1 outside -- nerfacc 0.3.3 `_update` drops those samples before it
 * evaluates them, and nsr_occupancy_update_values leaves the cell of a negative value untouched. */
int nsr_occupancy_density_values_sphere(const float *logit, const float *x_unit, float density_bias, float step_size,
                                        float *occ, uint32_t capacity, const int32_t *n_cells, void *stream);
/* ... with the occupancy statistic of the selected cells evaluated by the caller (occ_values [capacity] fp32; NeuS:
 * nsr_neus_occupancy_values) instead of derived from a density logit */
int nsr_occupancy_update_values(const float *occ_values, float ema_decay, float occ_thre, const uint32_t *cells,
                                const float *occs_old, float *occs_new, uint8_t *binary, float *threshold,
                                uint32_t n_total_cells, uint32_t capacity, const int32_t *n_cells, void *stream);

/* the Mip-NeRF 360 distortion loss the reference takes from torch_efficient_distloss
 * (flatten_eff_distloss(weights, points, intervals, ray_indices), systems/nerf.py:103-106, systems/neus.py:131-139).
 * forward: ray_loss[r] = sum_ij w_i w_j |m_i - m_j| + 1/3 sum_i w_i^2 dt_i over the samples of ray r (sorted along the
 * ray); backward: grad_weights[i] = d(sum_r ray_loss[r]) / d w_i.  The mean over rays and the chain factor are the
 * caller's (torch_efficient_distloss/__init__.py). */
int nsr_distortion_loss_forward(const int32_t *packed_info, const float *weights, const float *midpoints,
                                const float *intervals, float *ray_loss, uint32_t n_rays, void *stream);
int nsr_distortion_loss_backward(const int32_t *packed_info, const float *weights, const float *midpoints,
                                 const float *intervals, float *grad_weights, uint32_t n_rays, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Fused training-step glue, see csrc/fused.hip
 * ------------------------------------------------------------------------------------------------ */
/* x01 = contract_to_unisphere(o[r] + d[r]*(t0+t1)/2); dirs_out (may be NULL) = d[r]   (nerf.py:66-69,95-99) */
int nsr_sample_positions_unit(const float *rays_o, const float *rays_d, const int64_t *ray_indices,
                              const float *t_starts, const float *t_ends, float radius, int contraction, float *x01,
                              float *dirs_out, uint32_t n, const int32_t *n_dev, void *stream);
/* kept_counts[r] = number of leading samples of ray r with transmittance >= early_stop_eps, where alpha comes from
 * trunc_exp(mlp_out[:,0] + density_bias) -- nerfacc's render_visibility with alpha_thre == 0 (nerf.py:82-93) */
int nsr_visibility_prefix(const nsr_half *mlp_out, uint32_t stride, float density_bias, const float *t_starts,
                          const float *t_ends, const int32_t *packed_info, float early_stop_eps, int32_t *kept_counts,
                          uint32_t n_rays, void *stream);
/* Same compaction for up to 8 per-sample row arrays (host arrays of device pointers; row_bytes multiples of 4): the
 * kept prefix of a ray is contiguous before and after pruning, so this is a per-ray memcpy.  Lets the main pass
 * reuse the encodings / MLP activations the sigma pass already computed.  dirs_out / ray_indices_out may be NULL. */
int nsr_copy_ray_prefix_rows(const int32_t *packed_old, const int32_t *packed_new, uint32_t n_arrays,
                             const void *const *src, void *const *dst, const uint32_t *row_bytes, const float *rays_d,
                             float *dirs_out, int64_t *ray_indices_out, uint32_t n_rays, void *stream);
/* The fused step's instance of that copy (F = 2 level-major encoding, 16-half feature rows, n_hidden <= 2 saved
 * activation rows), one lane per kept sample with every row's load issued before the first store; *_capacity are the row
 * strides of the level-major / per-layer arrays in the marched and the kept layout.  Also writes ray_indices and tex_in */
int nsr_nerf_copy_kept_rows(const int32_t *packed_marched, const int32_t *packed_kept, const float *t_starts,
                            const float *t_ends, const float *x01, const nsr_half *enc, const nsr_half *out1,
                            const nsr_half *acts1, float *t_starts_out, float *t_ends_out, float *x01_out,
                            nsr_half *enc_out, nsr_half *out1_out, nsr_half *acts1_out, uint32_t n_levels,
                            uint32_t n_hidden, uint32_t marched_capacity, uint32_t kept_capacity, const float *rays_d,
                            int64_t *ray_indices_out, nsr_half *tex_in, uint32_t n_rays, void *stream);
/* Round 5 -- the packing folded into that copy: nsr_visibility_prefix_sums leaves, beside the kept counts, one sum per block
 * of 8 rays (block_sums: nsr_div_up(n_rays, 8) words, 16-byte aligned); every wave of the copy then forms its ray's offset
 * itself and WRITES packed_kept [n_rays,2], total_kept[1] and the statistics of nsr_pack_from_counts_capped (stats may be
 * NULL; rays past kept_capacity samples are truncated as there): one launch instead of scan + copy in the step's chain. */
int nsr_visibility_prefix_sums(const nsr_half *mlp_out, uint32_t stride, float density_bias, const float *t_starts,
                               const float *t_ends, const int32_t *packed_info, float early_stop_eps, int32_t *kept_counts,
                               int32_t *block_sums, uint32_t n_rays, void *stream);
int nsr_nerf_copy_kept_rows_scan(const int32_t *packed_marched, const int32_t *kept_counts, const int32_t *block_sums,
                                 int32_t *packed_kept, int32_t *total_kept, int32_t *stats, const float *t_starts,
                                 const float *t_ends, const float *x01, const nsr_half *enc, const nsr_half *out1,
                                 const nsr_half *acts1, float *t_starts_out, float *t_ends_out, float *x01_out,
                                 nsr_half *enc_out, nsr_half *out1_out, nsr_half *acts1_out, uint32_t n_levels,
                                 uint32_t n_hidden, uint32_t marched_capacity, uint32_t kept_capacity, const float *rays_d,
                                 int64_t *ray_indices_out, nsr_half *tex_in, uint32_t n_rays, void *stream);
/* tex_in[n,32] (half) = [mlp_out[:, :16] | SH4((dirs+1)/2)]   (texture.py:24-26) */
int nsr_texture_input(const nsr_half *mlp_out, uint32_t stride, const float *dirs, nsr_half *tex_in, uint32_t n,
                      const int32_t *n_dev, void *stream);
/* density = exp(mlp_out[:,0] + bias); weights/trans [n]; comp_rgb[R,3] = sum w*rgb + background*(1-opacity)
 * (nerf.py:105-109); rgb: half rows of rgb_stride, first 3 columns */
int nsr_composite_forward(const nsr_half *mlp_out, uint32_t stride, float density_bias, const float *t_starts,
                          const float *t_ends, const nsr_half *rgb, uint32_t rgb_stride, const int32_t *packed_info,
                          const float *background, float *weights, float *trans, float *comp_rgb, float *opacity,
                          float *depth, uint32_t n_rays, void *stream);
int nsr_composite_backward(const nsr_half *mlp_out, uint32_t stride, float density_bias, const float *t_starts,
                           const float *t_ends, const nsr_half *rgb, uint32_t rgb_stride, const int32_t *packed_info,
                           const float *background, const float *weights, const float *trans,
                           const float *grad_comp_rgb, const float *grad_opacity, const float *grad_depth,
                           float *grad_rgb, float *grad_logit, uint32_t n_rays, void *stream);
/* composite backward with the gradient of the masked smooth-L1 loss (nsr_smooth_l1_valid_backward) evaluated inside */
int nsr_composite_backward_smooth_l1(const nsr_half *mlp_out, uint32_t stride, float density_bias, const float *t_starts,
                                     const float *t_ends, const nsr_half *rgb, uint32_t rgb_stride,
                                     const int32_t *packed_info, const float *background, const float *weights,
                                     const float *trans, const float *comp_rgb, const float *opacity, const float *gt_rgb,
                                     const float *acc2, float grad_scale, float *grad_rgb, float *grad_logit,
                                     uint32_t n_rays, void *stream);
/* The same forward / backward pair with the loss reduction folded in (csrc/step.hip uses it when one call runs both): the
 * forward leaves one (loss sum, valid rays) partial per block in `partials` (nsr_composite_l1_partials_floats(n_rays)
 * floats, no initialisation), every block of the backward sums them in a fixed order and block 0 writes acc2 -- no
 * atomics and no one-workgroup reduction kernel between the two launches. */
uint64_t nsr_composite_l1_partials_floats(uint32_t n_rays);
int nsr_composite_forward_smooth_l1(const nsr_half *mlp_out, uint32_t stride, float density_bias, const float *t_starts,
                                    const float *t_ends, const nsr_half *rgb, uint32_t rgb_stride,
                                    const int32_t *packed_info, const float *background, float *weights, float *trans,
                                    float *comp_rgb, float *opacity, float *depth, const float *gt_rgb, float *partials,
                                    uint32_t n_rays, void *stream);
int nsr_composite_backward_smooth_l1_partials(const nsr_half *mlp_out, uint32_t stride, float density_bias,
                                              const float *t_starts, const float *t_ends, const nsr_half *rgb,
                                              uint32_t rgb_stride, const int32_t *packed_info, const float *background,
                                              const float *weights, const float *trans, const float *comp_rgb,
                                              const float *opacity, const float *gt_rgb, const float *partials,
                                              float *acc2, float grad_scale, float *grad_rgb, float *grad_logit,
                                              uint32_t n_rays, void *stream);
 /* Sample-partitioned compositing (round 6; nerfacc.render_weight_from_density + accumulate_along_rays, reference
 * models/nerf.py:105-108, with trunc_exp + density bias of models/geometry.py:122-156 folded in): one lane per kept sample, a
 * wave per 64 samples of the packed arrays whatever the ray boundaries.  ray_indices[n_samples] (int64) names each sample's
 * ray, packed_info [n_rays][2] its (start, count) -- an exclusive scan over the rays.  n_samples = capacity of the sample
 * arrays, n_samples_dev (may be NULL) the live count.  partials (may be NULL): nsr_composite_l1_partials_floats(n_rays) floats
 * for the folded masked smooth-L1 (forward and backward of one step take the same n_rays / n_samples). */
int nsr_composite_forward_samples(const nsr_half *mlp_out, uint32_t stride, float density_bias, const float *t_starts,
                                  const float *t_ends, const nsr_half *rgb, uint32_t rgb_stride, const int32_t *packed_info,
                                  const int64_t *ray_indices, const float *background, float *weights, float *trans,
                                  float *comp_rgb, float *opacity, float *depth, const float *gt_rgb, float *partials,
                                  uint32_t n_rays, uint32_t n_samples, const int32_t *n_samples_dev, void *stream);
int nsr_composite_backward_samples(const nsr_half *mlp_out, uint32_t stride, float density_bias, const float *t_starts,
                                   const float *t_ends, const nsr_half *rgb, uint32_t rgb_stride, const int32_t *packed_info,
                                   const int64_t *ray_indices, const float *background, const float *weights,
                                   const float *trans, const float *grad_comp_rgb, const float *grad_opacity,
                                   const float *grad_depth, const float *grad_weights, const float *comp_rgb,
                                   const float *opacity, const float *gt_rgb, const float *partials, float *acc2,
                                   float grad_scale, float *grad_rgb, float *grad_logit, uint32_t n_rays, uint32_t n_samples,
                                   const int32_t *n_samples_dev, void *stream);
/* acc2[0] += sum of smooth_l1 over valid rays (opacity > 0) x 3 channels, acc2[1] += number of valid rays
 * (loss = acc2[0] / (3*acc2[1]), systems/nerf.py:97); backward writes grad_scale * dloss/dcomp_rgb */
int nsr_smooth_l1_valid(const float *comp_rgb, const float *opacity, const float *gt_rgb, float *acc2,
                        uint32_t n_rays, void *stream);
/* the same two sums WRITTEN to acc2 by one workgroup (no same-address atomics, no zeroing by the caller) */
int nsr_smooth_l1_valid_set(const float *comp_rgb, const float *opacity, const float *gt_rgb, float *acc2,
                            uint32_t n_rays, void *stream);
int nsr_smooth_l1_valid_backward(const float *comp_rgb, const float *opacity, const float *gt_rgb, const float *acc2,
                                 float grad_scale, float *grad_comp_rgb, uint32_t n_rays, void *stream);
/* training-ray gather (systems/nerf.py:38-79, models/ray_utils.py:23-43): rays[n,6], rgb[n,3], fg[n] */
int nsr_gather_train_rays(const float *images, const float *masks, const float *directions, const float *c2w,
                          const int64_t *index, const int64_t *px, const int64_t *py, const float *background,
                          int height, int width, int apply_mask, float *rays, float *rgb, float *fg, uint32_t n,
                          void *stream);

/* everything a training ray needs before marching in ONE launch: pixel choice from 4 uniform rows u01[4,n]
 * (image, x, y, jitter), pixel gather, get_rays + normalise, background blend, slab test against `aabb`
 * (device[6]) and the stratified jitter t_min += u*jitter_step (0 disables).  rays[n,6] and the split rays_o/rays_d
 * (systems/nerf.py:38-79, models/ray_utils.py:23-43, nerfacc ray_aabb_intersect + ray_marching stratified) */
int nsr_prepare_train_rays(const float *images, const float *masks, const float *directions, const float *c2w,
                           const float *u01, const float *background, int n_images, int height, int width,
                           int apply_mask, const float *aabb, float jitter_step, float *rays, float *rays_o,
                           float *rays_d, float *rgb, float *fg, float *t_min, float *t_max, uint32_t n,
                           const int32_t *n_active, void *stream);
/* n_active (device, may be NULL = all n): slots >= *n_active become DEAD rays (t_min = t_max = 1e10: no samples,
 * opacity 0, outside the loss), so the dynamic batch size of systems/nerf.py:93-95 can live on the device:
 *   t = int(n_rays * (target_samples / n_samples)); n_rays = min(int(n_rays * 0.9 + t * 0.1), max_rays)
 * in double precision (Python's arithmetic); n_samples <= 0 or target_samples == 0 leaves n_rays unchanged.
 * rays_accum (device int64[1], may be NULL) += the ray count of THIS batch (read before the update). */
int nsr_update_ray_count(const int32_t *n_samples, int32_t *n_rays, int32_t target_samples, int32_t max_rays,
                         int64_t *rays_accum, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Native orchestration of the fused NeRF training step (csrc/step.hip): one C call per PHASE issues all of its
 * launches back to back (from Python the main queue idles ~40 % of the step on interpreter/allocator overhead).
 * The caller sizes ONE workspace per phase with the *_layout() functions (host-only) and reads results at the
 * returned byte offsets.  Reference: models/nerf.py:61-127 + systems/nerf.py:97.
 * ------------------------------------------------------------------------------------------------ */
typedef struct NsrNerfStepDesc {
    NsrGridDesc grid;
    NsrMlpDesc mlp_density; /* encoding -> 16 features (column 0 = density logit) */
    NsrMlpDesc mlp_color;   /* [16 features | 16 SH] -> rgb (sigmoid) */
    float radius;           /* scene_aabb = [-radius, radius]^3 */
    int contraction;        /* NSR_CONTRACT_AABB */
    float density_bias;     /* density = exp(logit + bias), models/geometry.py:127 */
    float early_stop_eps;   /* nerfacc ray_marching default 1e-4 */
    float grad_scale;       /* fp16 backward-chain scale inside the MLP kernels (128) */
    float loss_scale;       /* multiplies dloss/dcomp_rgb (1) */
} NsrNerfStepDesc;

typedef struct NsrNerfPruneLayout { uint64_t x01, enc, out1, acts1, total_bytes; } NsrNerfPruneLayout;
typedef struct NsrNerfMainLayout {
    uint64_t ray_indices, t_starts, t_ends, weights, comp_rgb, opacity, depth, loss_acc; /* results the host reads */
    uint64_t trans, x01, dirs, enc, out1, acts1, tex_in, out2, acts2, g_comp, d_rgb, d_logit, d_tex, d_enc, partials,
        grid_ws;                                                                           /* internals */
    uint64_t total_bytes;
} NsrNerfMainLayout;

/* opt-in HIP-event timing of the heavy launches inside the two passes (events on the launch stream; zero cost when
 * off).  collect(tag) synchronises on the recorded events and sums them; collect(-1) resets. */
#define NSR_PROF_GRID_FORWARD 0
#define NSR_PROF_GRID_BACKWARD 1
#define NSR_PROF_MLP_FORWARD_DENSITY 2
#define NSR_PROF_MLP_FORWARD_COLOR 3
#define NSR_PROF_MLP_BACKWARD_COLOR 4
#define NSR_PROF_MLP_BACKWARD_DENSITY 5
#define NSR_PROF_GRID_BACKWARD_BIN 6 /* item binning of the table backward, on the main pass's helper stream */
/* Forms of the step's kernels that stay switchable for same-process A/B runs (bench.py `step_forms_ab`) and as a fallback:
 * key 0: nsr_mlp_dgrad_pair instead of two data-gradient launches; key 2: sample-partitioned compositing
 * (nsr_composite_*_samples) instead of one wave per ray; key 5: the pass's fork events ride on the kernels in front of them
 * (hipExtLaunchKernelGGL stop event) instead of being recorded behind them; key 9: the value is the block cap of the pass's
 * own weight-gradient launches (nsr_mlp_wgrad_max_blocks around them only; default 128, 0 = leave the library's).
 * value < 0 queries; returns the previous value (-1: unknown key). */
int nsr_nerf_step_variant(int key, int value);
/* One-shot: the NEXT pruning pass makes its stream wait for that hipEvent_t between its hash encode and its density MLP (the
 * first kernel that reads network weights) -- for a caller that defers its join with the weight-gradient kernels
 * (nsr_nerf_defer_wgrad_join) and hands over the event of its optimizer launch for the network weights, instead of the step's
 * stream waiting in front of the encode.  owner: the step descriptor that pass will be called with (a pass of any other
 * descriptor leaves the wait armed); event == NULL clears (only the owner's own).  The caller must wait for the event itself
 * before anything else reads the weights, and clear it before the event is destroyed. */
int nsr_nerf_wait_before_mlp(const NsrNerfStepDesc *owner, void *event);
/* the stream the main pass runs its overlapped work on (item binning, weight-gradient kernels); created on first use */
void *nsr_nerf_helper_stream(void);
/* `stream` waits for the point of the last main pass where its kept rows exist (behind nsr_nerf_main_pass*'s first kernel) */
int nsr_nerf_wait_kept_rows(void *stream);
/* on != 0: the main pass does not join its weight-gradient kernels itself -- the caller queues its optimizer launch behind
 * them on nsr_nerf_helper_stream() and makes the step's stream wait for that.  Returns the previous setting. */
int nsr_nerf_defer_wgrad_join(int on);
void nsr_profile_enable(int on);
int nsr_profile_collect(int tag, double *total_ms, uint64_t *launches, uint64_t *units);

int nsr_nerf_prune_layout(const NsrNerfStepDesc *d, uint32_t n_marched, NsrNerfPruneLayout *out);
/* sigma pass over all marched samples -> kept_counts[n_rays], packed_kept[n_rays,2], total_kept[1] (device).
 * n_marched_dev (may be NULL): device-side number of marched samples, n_marched then being the buffer capacity;
 * kept_capacity / kept_stats: see nsr_pack_from_counts_capped (0 / NULL: unlimited, no statistics). */
int nsr_nerf_prune_pass(const NsrNerfStepDesc *d, const float *rays_o, const float *rays_d, const int64_t *ray_indices,
                        const float *t_starts, const float *t_ends, const int32_t *packed_info, const nsr_half *table,
                        const nsr_half *w_density, void *workspace, int32_t *kept_counts, int32_t *packed_kept,
                        int32_t *total_kept, uint32_t n_marched, uint32_t n_rays, const int32_t *n_marched_dev,
                        uint32_t kept_capacity, int32_t *kept_stats, const float *x01_marched, void *stream);
/* Round 5 -- the same pass for a caller that queues nsr_nerf_main_pass* / nsr_nerf_render_forward on the SAME stream with
 * the SAME packed_kept / total_kept right behind it: packed_kept, total_kept and kept_stats are then written by that pass's
 * first kernel (nsr_nerf_copy_kept_rows_scan) and nothing may read them in between; the one-workgroup scan leaves the
 * step's chain.  Falls back to the plain pass for shapes that copy does not cover. */
int nsr_nerf_prune_pass_deferred(const NsrNerfStepDesc *d, const float *rays_o, const float *rays_d,
                                 const int64_t *ray_indices, const float *t_starts, const float *t_ends,
                                 const int32_t *packed_info, const nsr_half *table, const nsr_half *w_density,
                                 void *workspace, int32_t *kept_counts, int32_t *packed_kept, int32_t *total_kept,
                                 uint32_t n_marched, uint32_t n_rays, const int32_t *n_marched_dev,
                                 uint32_t kept_capacity, int32_t *kept_stats, const float *x01_marched, void *stream);
/* x01_marched (may be NULL; both passes): unit-cube positions [n_marched, 3] of the marched samples computed by the caller
 * ahead of the step (nsr_sample_positions_unit on its marching stream); NULL: formed inside, in the workspace */
int nsr_nerf_main_layout(const NsrNerfStepDesc *d, uint32_t n_kept, uint32_t n_rays, NsrNerfMainLayout *out);
/* forward + loss (+ backward when compute_grads): gradients are ADDED to grad_density_mlp / grad_color_mlp and
 * OVERWRITE grad_table; t_starts/t_ends/packed_marched describe the marched samples, packed_kept the kept ones.
 * n_kept_dev (may be NULL): device-side number of kept samples, n_kept (and n_marched) then being buffer capacities */
int nsr_nerf_main_pass(const NsrNerfStepDesc *d, const void *prune_workspace, uint32_t n_marched,
                       const int32_t *packed_marched, const int32_t *packed_kept, const float *t_starts,
                       const float *t_ends, const float *rays_d, const float *background, const float *gt_rgb,
                       const nsr_half *w_density, const nsr_half *w_color, float *grad_density_mlp, float *grad_table,
                       float *grad_color_mlp, void *workspace, uint32_t n_kept, uint32_t n_rays, int compute_grads,
                       const int32_t *n_kept_dev, const float *x01_marched, const struct NsrTableAdam *table_adam,
                       void *stream);
/* table_adam (may be NULL): apply AdamW to the hash table inside the table backward (NsrTableAdam below) -- grad_table is
 * then neither written nor read; with S == 0 kept samples the caller's optimizer still has to decay the table. */

/* The main pass split AT THE LOSS, for callers that own it: the reference's system computes its loss in torch on the
 * model's output dict and calls backward() (systems/nerf.py:87-99) -- nsr.models.FusedNeRFModel wraps these two in one
 * torch.autograd.Function.  _forward = everything of nsr_nerf_main_pass up to the composite (outputs and every saved
 * activation at the nsr_nerf_main_layout offsets of `workspace`; with prepare_backward != 0 the table-backward items are
 * binned on the helper stream meanwhile).  _backward = the rest, from the upstream gradients: comp_rgb[n_rays,3]
 * (required), opacity[n_rays], depth[n_rays], weights[n_kept] (each may be NULL = zero); gradients are ADDED to
 * grad_density_mlp / grad_color_mlp and OVERWRITE grad_table.  Same n_kept / n_rays / workspace in both calls, nothing else
 * may use the helper stream's bins in between (a no-grad forward passes prepare_backward = 0). */
typedef struct NsrRenderGrads {
    const float *comp_rgb, *opacity, *depth, *weights;
} NsrRenderGrads;
int nsr_nerf_render_forward(const NsrNerfStepDesc *d, const void *prune_workspace, uint32_t n_marched,
                            const int32_t *packed_marched, const int32_t *packed_kept, const float *t_starts,
                            const float *t_ends, const float *rays_d, const float *background, const nsr_half *w_density,
                            const nsr_half *w_color, void *workspace, uint32_t n_kept, uint32_t n_rays,
                            int prepare_backward, const int32_t *n_kept_dev, const float *x01_marched, void *stream);
int nsr_nerf_render_backward(const NsrNerfStepDesc *d, const void *prune_workspace, uint32_t n_marched,
                             const int32_t *packed_marched, const int32_t *packed_kept, const float *rays_d,
                             const float *background, const NsrRenderGrads *upstream, const nsr_half *w_density,
                             const nsr_half *w_color, float *grad_density_mlp, float *grad_table, float *grad_color_mlp,
                             void *workspace, uint32_t n_kept, uint32_t n_rays, const int32_t *n_kept_dev, void *stream);

/* The ray-sharded (multi-GPU) form of the main pass -- the reference gets its gradient exchange from Lightning DDP
 * (launch.py:93-107); here the step itself hands the gradients to the exchange in pieces, as they become final:
 *   * the table gradient leaves as bf16 in grad_bf16 (indexed like the table), in n_groups launches over the level runs
 *     [level_begin[g], level_end[g]) in the order given (finest levels first); event_group[g] (a hipEvent_t of the caller,
 *     may be NULL) is recorded on `stream` behind group g, so that a communication stream can reduce-scatter group g
 *     while group g + 1 is still accumulating;
 *   * event_small (may be NULL) is recorded behind the weight-gradient kernels of both MLPs (which finish before the
 *     table backward starts): the small fp32 gradients can be all-reduced underneath the table backward.
 * grad_table / table_adam must be NULL.  With S == 0 kept samples grad_bf16 is zero-filled and every event recorded. */
typedef struct NsrTableExchange {
    void *grad_bf16;
    uint64_t grad_bf16_elems; /* capacity of grad_bf16 (>= table parameters; the tail is padding of the exchange) */
    uint32_t n_groups;        /* 1..4 */
    uint32_t level_begin[4], level_end[4];
    void *event_group[4];
    void *event_small;
} NsrTableExchange;
int nsr_nerf_main_pass_exchange(const NsrNerfStepDesc *d, const void *prune_workspace, uint32_t n_marched,
                                const int32_t *packed_marched, const int32_t *packed_kept, const float *t_starts,
                                const float *t_ends, const float *rays_d, const float *background, const float *gt_rgb,
                                const nsr_half *w_density, const nsr_half *w_color, float *grad_density_mlp,
                                float *grad_color_mlp, void *workspace, uint32_t n_kept, uint32_t n_rays,
                                const int32_t *n_kept_dev, const float *x01_marched, const NsrTableExchange *exchange,
                                void *stream);

/* ------------------------------------------------------------------------------------------------
 * SURVEY.md section 8f "next" row 1: fused AdamW over the flat fp32 params (configs/<name>.yaml optimizer:
 * AdamW lr 0.01 betas (0.9,0.99) eps 1e-15, systems/utils.py:314-325) that also refreshes the fp16
 * shadow the kernels read and zeroes the gradient.
 * ------------------------------------------------------------------------------------------------ */
int nsr_adamw_step(float *params, float *grad, float *exp_avg, float *exp_avg_sq, nsr_half *shadow_half,
                   uint64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                   float bias_correction1, float bias_correction2, float grad_unscale, int zero_grad, const float *hyper,
                   uint64_t zero_first_n, void *stream);
/* zero_first_n (with zero_grad != 0): only grad[0 .. zero_first_n) is zeroed (0 = all of it; a multiple of 4) -- the fused
 * step OVERWRITES the hash-table part of the gradient every step, zeroing those 50 MB again is wasted bandwidth */
/* torch.optim.AdamW over up to 32 small tensors in ONE launch, each with its own learning rate (parameter groups of the
 * NeuS systems: fp32 heads 0.01, variance 0.001 -- configs/neus-*.yaml optimizer.params); zero_grad != 0 clears grad */
typedef struct NsrAdamSegment {
    float *params, *grad, *exp_avg, *exp_avg_sq;
    uint64_t n;
    float lr;
} NsrAdamSegment;
/* step_dev / hyper_dev (both or neither; int32[1] / float[4] on the device): the optimizer's step count lives on the device --
 * a one-thread launch ahead of the update advances it and leaves 1 - beta^step in hyper_dev[1..2] (bias_correction1/2 are then
 * ignored) UNLESS a registered overflow guard (nsr_overflow_guard) has flagged the step: then nothing is touched, as
 * GradScaler.step() skips optimizer.step().  That launch also performs GradScaler.update() on the guard's state: call this
 * entry LAST among a step's optimizer launches. */
int nsr_adamw_multi(const NsrAdamSegment *segments, uint32_t n_segments, float beta1, float beta2, float eps,
                    float weight_decay, float bias_correction1, float bias_correction2, int zero_grad, int32_t *step_dev,
                    float *hyper_dev, void *stream);
/* Stencil mode of the owner-computes table backward (finite-difference normals, reference models/geometry.py:181-199):
 * x7 = positions [7][n_centre][3] (sample, then the six +-eps taps), dy_level_major = [L][7 n_centre][F] fp32.  A tap that
 * stays in its sample's cell moves one coordinate inside a trilinear cell, so it is folded exactly into the sample's items
 * (w G0 + sum_a dw/dx_a D_a); only taps that cross into a neighbouring cell keep items of their own.  workspace as for the
 * plain call over 7 n_centre points, tap_workspace: nsr_hashgrid_backward_params_taps_workspace_floats floats. */
uint64_t nsr_hashgrid_backward_params_taps_workspace_floats(const NsrGridDesc *desc, uint32_t n_centre);
int nsr_hashgrid_backward_params_owner_bin_taps(const float *x7, float *workspace, float *tap_workspace, uint32_t n_centre,
                                                uint32_t level_mask_count, const NsrGridDesc *desc, void *stream);
/* ... when nsr_hashgrid_forward_taps_masks (same x7, n_centre, level_mask_count) already wrote the crossing masks */
int nsr_hashgrid_backward_params_owner_bin_taps_masked(const float *x7, float *workspace, float *tap_workspace,
                                                       uint32_t n_centre, uint32_t level_mask_count,
                                                       const NsrGridDesc *desc, void *stream);
int nsr_hashgrid_backward_params_owner_accumulate_taps(const float *x7, const float *dy_level_major, float *grad_table,
                                                       float *workspace, float *tap_workspace, uint32_t n_centre,
                                                       uint32_t level_mask_count, int accumulate,
                                                       const NsrGridDesc *desc, void *stream);
/* AdamW fused into the owner-computes table backward: the workgroup that owns a table slice applies the update to it
 * from the gradient it holds in LDS (no gradient store + separate optimizer read).  params / exp_avg / exp_avg_sq /
 * shadow point at the TABLE part of the flat parameter vector (entry 0 of level 0 first; 16-byte aligned, shadow 8);
 * step / hyper: the device schedule state of nsr_adam_tick / nsr_adamw_step_scheduled -- read here, NOT advanced: the
 * caller advances it afterwards with nsr_adamw_step_scheduled over the remaining (MLP) parameters.  Bit-identical to
 * nsr_hashgrid_backward_params_owner_accumulate + nsr_adamw_step on the same tensors. */
/* Overflow guard of the fused fp16 training step -- Lightning's `precision: 16` (reference configs/nerf-blender.yaml:103):
 * torch.cuda.amp.GradScaler skips the optimizer step when a gradient is inf / NaN, halves the loss scale and doubles it again
 * after growth_interval clean steps.  state: int32[8] in device memory {found-inf flag of even steps, of odd steps, scale
 * (float bits), clean steps, skipped steps, growth interval, -, -}.  Registered (state != NULL) around the launches of ONE
 * trainer's step and withdrawn (NULL) afterwards -- it is read on the host when a launch is queued: nsr_mlp_dgrad_pair scales
 * dL/dy by state[2] instead of its grad_scale argument and raises this step's flag on a non-finite encoding gradient, the
 * table backward's fused AdamW (NsrTableAdam) and nsr_adamw_step_scheduled* leave weights, moments and fp16 images untouched
 * when it is set, and the latter updates the scale.  scale0 = the constant the weight-gradient reductions unscale by (the step
 * descriptor's grad_scale).  The NeuS step: nsr_neus_composite_backward* (the first kernel of its backward) starts the step --
 * it raises the flag on a non-finite loss gradient, nsr_neus_shade_backward* on a non-finite gradient at the SDF network's
 * output (an overflow of the fp16 colour network lands there) -- nsr_adam_tick, nsr_adamw_step and nsr_adamw_multi skip when it
 * is set, and nsr_adamw_multi (with a device-side step count) updates the scale.  Returns the current parity. */
int nsr_overflow_guard(int32_t *state, float scale0);

typedef struct NsrTableAdam {
    float *params, *exp_avg, *exp_avg_sq;
    nsr_half *shadow; /* may be NULL */
    const int32_t *step;
    const float *hyper;
    double base_lr, beta1, beta2, gamma;
    int32_t milestone0, milestone1, milestone2;
    float eps, weight_decay;
} NsrTableAdam;
int nsr_hashgrid_backward_params_owner_accumulate_adam(const float *x, const void *dy, int dy_layout, uint32_t dy_stride,
                                                       float *workspace, uint32_t n, uint32_t level_mask_count,
                                                       float grad_scale, const NsrGridDesc *desc, const int32_t *n_dev,
                                                       const NsrTableAdam *adam, void *stream);


/* The same write-out for the two other accumulation modes (the fused NeuS steps, nsr/fused_neus.py): first + second order
 * in one pass (analytic normals; items already binned when binned != 0) and the finite-difference stencil mode. */
int nsr_hashgrid_backward_params_owner_with_second_order_adam(const float *x, const float *dy_first_lm, const float *dy,
                                                              uint32_t dy_stride, const float *g, float *workspace,
                                                              uint32_t n, uint32_t level_mask_count, int binned,
                                                              const NsrGridDesc *desc, const NsrTableAdam *adam,
                                                              void *stream);
int nsr_hashgrid_backward_params_owner_accumulate_taps_adam(const float *x7, const float *dy_level_major, float *workspace,
                                                            float *tap_workspace, uint32_t n_centre,
                                                            uint32_t level_mask_count, const NsrGridDesc *desc,
                                                            const NsrTableAdam *adam, void *stream);
/* ... and with the gradient written as bf16 (round to nearest even) into a caller buffer [n_entries * F] -- the send buffer of
 * the multi-GPU exchange (nsr/parallel.py:ShardedAdamW) -- instead of an fp32 gradient: written once, every entry. */
int nsr_hashgrid_backward_params_owner_with_second_order_bf16(const float *x, const float *dy_first_lm, const float *dy,
                                                              uint32_t dy_stride, const float *g, uint16_t *grad_bf16,
                                                              float *workspace, uint32_t n, uint32_t level_mask_count,
                                                              int binned, const NsrGridDesc *desc, void *stream);
int nsr_hashgrid_backward_params_owner_accumulate_taps_bf16(const float *x7, const float *dy_level_major,
                                                            uint16_t *grad_bf16, float *workspace, float *tap_workspace,
                                                            uint32_t n_centre, uint32_t level_mask_count,
                                                            const NsrGridDesc *desc, void *stream);
/* The optimizer step of the asynchronous trainer in ONE launch: nsr_adam_tick + nsr_adamw_step over up to two tensors
 * (a: hash table + density MLP with its partial re-zeroing, b: colour MLP; n_b == 0: one tensor).  hyper12: 12 floats, 8-byte
 * aligned, zero-initialised ([0..7] as for nsr_adam_tick, [8] ticket counter).  Bit-identical to the separate launches. */
int nsr_adamw_step_scheduled(float *params_a, float *grad_a, float *exp_avg_a, float *exp_avg_sq_a, nsr_half *shadow_a,
                             uint64_t n_a, uint64_t zero_first_n_a, float *params_b, float *grad_b, float *exp_avg_b,
                             float *exp_avg_sq_b, nsr_half *shadow_b, uint64_t n_b, int32_t *step, float *hyper12,
                             double base_lr, double beta1, double beta2, double gamma, int32_t milestone0,
                             int32_t milestone1, int32_t milestone2, float eps, float weight_decay, float grad_unscale,
                             int zero_grad, void *stream);
/* The same launch with the advanced schedule state written to (step_out, hyper12_out) instead of in place -- the other half
 * of a double buffer: a kernel on another stream (the table backward's fused AdamW, NsrTableAdam) may read (step, hyper12)
 * while this launch runs.  The ticket word stays in hyper12[8].  step_out == step && hyper12_out == hyper12: in place. */
int nsr_adamw_step_scheduled_to(float *params_a, float *grad_a, float *exp_avg_a, float *exp_avg_sq_a, nsr_half *shadow_a,
                                uint64_t n_a, uint64_t zero_first_n_a, float *params_b, float *grad_b, float *exp_avg_b,
                                float *exp_avg_sq_b, nsr_half *shadow_b, uint64_t n_b, int32_t *step, float *hyper12,
                                int32_t *step_out, float *hyper12_out, double base_lr, double beta1, double beta2,
                                double gamma, int32_t milestone0, int32_t milestone1, int32_t milestone2, float eps,
                                float weight_decay, float grad_unscale, int zero_grad, void *stream);
/* the one collective of the path is the mean all-reduce of the gradients; the 50 MB table
 * gradient travels as fp16 (dst = half(src * scale) before, dst = float(src) * scale after; nsr/parallel.py) */
int nsr_scale_to_half(const float *src, nsr_half *dst, uint64_t n, float scale, void *stream);
int nsr_scale_from_half(const nsr_half *src, float *dst, uint64_t n, float scale, void *stream);
/* hyper (device, may be NULL): {lr, bias_correction1, bias_correction2} read on the device instead of the scalar
 * arguments; nsr_adam_tick needs 8 floats there, 8-byte aligned ([3..7]: its running beta powers).  nsr_adam_tick advances the device-side step counter (int32[1]) and writes them -- MultiStepLR
 * (configs nerf-blender.yaml:80-85: up to three milestones, pass INT32_MAX for unused ones) over base_lr, in double
 * arithmetic -- so that a captured (hipGraph) step needs no per-step host scalar. */
int nsr_adam_tick(int32_t *step, float *hyper, double base_lr, double beta1, double beta2, double gamma,
                  int32_t milestone0, int32_t milestone1, int32_t milestone2, void *stream);

/* ------------------------------------------------------------------------------------------------
 * fp32 "VanillaMLP" (reference models/network_utils.py:95-139: nn.Linear stack with biases, 64 neurons, ReLU or
 * Softplus(beta=100); weight norm folded by the caller) on f32 MFMA: the SDF network of models/geometry.py:146-150 and the
 * fp32 texture / background heads of configs/neus-dtu.yaml, configs/neuralangelo-dtu-wmask.yaml (csrc/vmlp.hip).
 * Parameter blob (fp32): W0[64][in_pad] b0[64] | (W1[64][64] b1[64]) | Wl[16][64] bl[16]; rows >= n_out of Wl / bl zero.
 * ------------------------------------------------------------------------------------------------ */
typedef struct NsrVmlpDesc {
    uint32_t n_in;       /* logical inputs (<= in_pad; the padding columns of W0 must be zero) */
    uint32_t in_pad;     /* 24, 32, 36 or 40 */
    uint32_t n_out;      /* <= 16 */
    uint32_t n_hidden;   /* 1 or 2 hidden layers of width 64 */
    uint32_t activation; /* 0 = ReLU, 1 = Softplus(beta=100, threshold=20) */
} NsrVmlpDesc;
uint64_t nsr_vmlp_blob_floats(const NsrVmlpDesc *desc);
uint64_t nsr_vmlp_backward_workspace_floats(const NsrVmlpDesc *desc, uint32_t n);
/* The nn.Linear tensors of a reference VanillaMLP layer (models/network_utils.py:95-139): weight_v [n_out][n_in] with
 * weight_g [n_out] (old-style torch weight_norm, W[r] = g[r] v[r] / |v[r]|) or the plain weight in weight_v with weight_g
 * NULL; grad_* receive the gradients (unfold).  nsr_vmlp_fold builds the padded parameter blob from n_hidden + 1 layers,
 * nsr_vmlp_unfold_gradient turns the blob's gradient into the gradients of those tensors (accumulate != 0: added). */
typedef struct NsrVanillaLayer {
    const float *weight_v, *weight_g, *bias;
    float *grad_v, *grad_g, *grad_bias;
    uint32_t n_out, n_in;
} NsrVanillaLayer;
int nsr_vmlp_fold(const NsrVmlpDesc *desc, const NsrVanillaLayer *layers, uint32_t n_layers, float *blob, void *stream);
int nsr_vmlp_unfold_gradient(const NsrVmlpDesc *desc, const NsrVanillaLayer *layers, uint32_t n_layers,
                             const float *grad_blob, int accumulate, void *stream);
/* x: fp32 rows [n][x_stride]; with enc != NULL the input is [2 x - 1 (3 columns of x) | enc (fp16 rows, n_in - 3 columns)]
 * (CompositeEncoding with include_xyz, models/network_utils.py:75-76); enc_stride = 0x80000000 | F selects the level-major
 * encoding [(n_in - 3) / F][n][F] that the fused encode kernels write, enc_stride = 0x40000000 | F the tile-major one
 * [ceil(n/16)][(n_in - 3) / F][16][F] (nsr_hashgrid_forward_ex, layout 2).  Rows < n_full write all 16 output columns to
 * out[n_full][16], rows >= n_full only column 0 to out_col0[n - n_full] (finite-difference taps).  g_in (may be NULL,
 * one hidden layer): [n][in_pad] = d out[0] / d input (analytic normal, models/geometry.py:176-180). */
int nsr_vmlp_forward(const NsrVmlpDesc *desc, const float *blob, const float *x, uint32_t x_stride, const nsr_half *enc,
                     uint32_t enc_stride, float *out, float *out_col0, float *g_in, uint32_t n, uint32_t n_full,
                     const int32_t *n_dev, void *stream);
/* d_out [n_full][16] / d_out_col0 [n - n_full]: gradients w.r.t. the outputs written by the forward; p_in (may be NULL):
 * [n][in_pad] = dL/d g_in (second-order terms of the analytic normal).  d_x (may be NULL): gradient w.r.t. input columns
 * [dx_first, dx_first + dx_count) (dx_count 0 = all), row-major [n][dx_stride] or, with dx_level_major_features = F,
 * level-major [dx_count/F][n][F] (what the owner-computes hash-grid backward reads).  grad_blob: blob layout, overwritten
 * (accumulate = 0) or added to.  partials: nsr_vmlp_backward_workspace_floats() floats. */
int nsr_vmlp_backward(const NsrVmlpDesc *desc, const float *blob, const float *x, uint32_t x_stride, const nsr_half *enc,
                      uint32_t enc_stride, const float *d_out, const float *d_out_col0, const float *p_in, float *d_x,
                      uint32_t dx_stride, uint32_t dx_first, uint32_t dx_count, uint32_t dx_level_major_features,
                      float *grad_blob, int accumulate, float *partials, uint32_t n, uint32_t n_full,
                      const int32_t *n_dev, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Fused NeuS step glue (csrc/neus.hip): reference models/neus.py:205-287, models/geometry.py:158-210,
 * models/neus.py:117-139, models/texture.py:23-30, loss terms of systems/neus.py:96-130.
 * acc: float[16] loss sums {L1, MSE, valid rays, mask BCE, opaque BCE, eikonal, sparsity, |laplace|, d/d inv_s, rays}
 * (zeroed by the caller before the forward pass).  loss_weights8: {lambda_rgb_l1, lambda_rgb_mse, lambda_mask,
 * lambda_opaque, lambda_eikonal, lambda_sparsity, lambda_curvature, sparsity_scale}.
 * ------------------------------------------------------------------------------------------------ */
/* x7: [1 + 6*taps][n][3] unit coordinates: the sample itself, then (taps != 0) the +-eps taps of models/geometry.py:182-194 */
int nsr_neus_points(const float *rays_o, const float *rays_d, const int64_t *ray_indices, const float *t_starts,
                    const float *t_ends, float radius, float eps, int taps, float *x7, float *dirs, uint32_t n,
                    const int32_t *n_dev, void *stream);
/* analytic (tap_sdf == NULL): grad = (2 g_in[:, 0:3] + dx01) / 2r; finite differences: tap_sdf [6][n], laplace out.
 * tex_in: [n][32] = [feature (n_feat) | SH4(dir) | normal | pad], fp16 (pad 1.0, fused colour MLP) or fp32 (pad 0) */
int nsr_neus_shade_forward(const float *sdf_out, const float *g_in, uint32_t g_stride, const float *dx01,
                           const float *tap_sdf, float eps, float radius, const float *dirs, const float *t_starts,
                           const float *t_ends, const float *inv_s, float cos_anneal_ratio, uint32_t n_feat,
                           float sparsity_scale, float *grad, float *normal, float *alpha, float *laplace, void *tex_in,
                           int tex_is_f32, float *acc, uint32_t n, const int32_t *n_dev, void *stream);
/* rgb_raw: [n][16] colour logits (fp16 or fp32), sigmoid (color_activation) applied here.  background: one colour
 * (background_stride 0) or, with the learned background, the per-ray comp_rgb_bg [n_rays][3] (background_stride 3):
 * comp_rgb_full = comp_rgb + background (1 - opacity)   (models/neus.py:273-283) */
int nsr_neus_composite_forward(const int32_t *packed_info, const float *alpha, const void *rgb_raw, int rgb_is_f32,
                               const float *normal, const float *t_starts, const float *t_ends, const float *background,
                               uint32_t background_stride, float *weights, float *trans, float *comp_rgb, float *opacity,
                               float *depth, float *comp_normal, float *comp_rgb_full, uint32_t n_rays, void *stream);
/* opacity_bg (may be NULL): rays_valid_full = opacity > 0 | opacity_bg > 0 */
int nsr_neus_loss_rays(const float *comp_rgb_full, const float *opacity, const float *opacity_bg, const float *gt_rgb,
                       const float *fg_mask, float *acc, uint32_t n_rays, const int32_t *n_active, void *stream);
/* d_background (may be NULL): dL / d comp_rgb_bg [n_rays][3] = dL/d comp_rgb_full (1 - opacity) */
int nsr_neus_composite_backward(const int32_t *packed_info, const float *alpha, const void *rgb_raw, int rgb_is_f32,
                                const float *weights, const float *trans, const float *background,
                                uint32_t background_stride, const float *opacity_bg, const float *comp_rgb_full,
                                const float *opacity, const float *gt_rgb, const float *fg_mask, const float *acc,
                                const float *loss_weights8, float loss_scale, float *d_alpha, float *d_rgb_raw,
                                float *d_background, uint32_t n_rays, const int32_t *n_active, void *stream);
/* d_out [n][16]: gradient w.r.t. the SDF network output; analytic: gx [n][3] = dL/d(dx01) (seeds the hash grid's double
 * backward) and p_in[:, 0:3] = dL/d g_in[:, 0:3]; finite differences: d_taps [6][n] */
int nsr_neus_shade_backward(const float *sdf_out, const float *grad, const float *normal, const float *dirs,
                            const float *t_starts, const float *t_ends, const float *inv_s, float cos_anneal_ratio,
                            const float *laplace, float eps, float radius, const float *d_alpha, const float *d_tex_in,
                            uint32_t n_feat, const float *loss_weights8, float loss_scale, float n_samples, float *d_out,
                            float *gx, float *p_in, uint32_t p_stride, float *d_taps, float *acc, uint32_t n,
                            const int32_t *n_dev, void *stream);

/* The two backward kernels with the gradients of a CALLER-OWNED loss instead of the built-in terms (nsr.models.FusedNeuSModel:
 * the reference's system forms its loss in torch on the model's output dict and calls backward(), systems/neus.py:96-139).
 * Every pointer may be NULL (= zero): per ray comp_rgb_full [R][3], comp_rgb [R][3], opacity [R], depth [R]; per sample
 * weights [n], sdf_samples [n], sdf_grad_samples [n][3], sdf_laplace_samples [n].  Pass loss_weights8 = zeros with it. */
typedef struct NsrNeusUpstream {
    const float *comp_rgb_full, *comp_rgb, *opacity, *depth, *weights, *sdf_samples, *sdf_grad_samples, *sdf_laplace_samples;
} NsrNeusUpstream;
int nsr_neus_composite_backward_ex(const int32_t *packed_info, const float *alpha, const void *rgb_raw, int rgb_is_f32,
                                   const float *weights, const float *trans, const float *background,
                                   uint32_t background_stride, const float *opacity_bg, const float *comp_rgb_full,
                                   const float *opacity, const float *gt_rgb, const float *fg_mask, const float *acc,
                                   const float *loss_weights8, float loss_scale, float *d_alpha, float *d_rgb_raw,
                                   float *d_background, uint32_t n_rays, const int32_t *n_active,
                                   const NsrNeusUpstream *upstream, const float *t_starts, const float *t_ends, void *stream);
int nsr_neus_shade_backward_ex(const float *sdf_out, const float *grad, const float *normal, const float *dirs,
                               const float *t_starts, const float *t_ends, const float *inv_s, float cos_anneal_ratio,
                               const float *laplace, float eps, float radius, const float *d_alpha, const float *d_tex_in,
                               uint32_t n_feat, const float *loss_weights8, float loss_scale, float n_samples, float *d_out,
                               float *gx, float *p_in, uint32_t p_stride, float *d_taps, float *acc, uint32_t n,
                               const int32_t *n_dev, const NsrNeusUpstream *upstream, void *stream);

/* occ[i] = clip((sigmoid((sdf + h) inv_s) - sigmoid((sdf - h) inv_s) + 1e-5) / (sigmoid((sdf + h) inv_s) + 1e-5), 0, 1),
 * h = step_size / 2, sdf = sdf_out[i][0] (rows of 16 floats), inv_s clipped to [1e-6, 1e6]   (models/neus.py:90-101) */
int nsr_neus_occupancy_values(const float *sdf_out, const float *inv_s, float step_size, float *occ, uint32_t n,
                              const int32_t *n_dev, void *stream);
/* inv_s[0] = exp(10 variance[0]) (models/neus.py:27-32); grad_variance (+)= acc[inv_s gradient slot] * inv_s * 10 */
int nsr_neus_inv_s(const float *variance, float *inv_s, void *stream);
int nsr_neus_variance_gradient(const float *acc, const float *inv_s, float *grad_variance, int accumulate, void *stream);

/* ---- NeRF++ background of the NeuS model (reference models/neus.py:169-203 `forward_bg_`; VolumeDensity with an fp32
 * VanillaMLP head models/geometry.py:116-130, trunc_exp models/utils.py:55-66, VolumeRadiance models/texture.py:23-30).
 * out16: [n][16] fp32 output rows of the density network (csrc/vmlp.hip), column 0 = density logit ---- */
/* kept_counts[r] = leading samples of ray r with transmittance >= early_stop_eps (ray_marching's sigma_fn pruning) */
int nsr_bg_visibility_prefix(const float *out16, float density_bias, const float *t_starts, const float *t_ends,
                             const int32_t *packed_info, float early_stop_eps, int32_t *kept_counts, uint32_t n_rays,
                             void *stream);
/* tex_in[i] = [out16[i][0:n_feat] | SH4(rays_d[ray_indices[i]]) rounded to fp16 | 0 ...] (fp32 rows of `stride`) */
int nsr_bg_texture_input(const float *out16, uint32_t n_feat, const float *rays_d, const int64_t *ray_indices,
                         float *tex_in, uint32_t stride, uint32_t n, const int32_t *n_dev, void *stream);
/* render_weight_from_density + accumulate_along_rays; comp_rgb includes background (1 - opacity) */
int nsr_bg_composite_forward(const int32_t *packed_info, const float *out16, float density_bias, const float *rgb_raw,
                             const float *t_starts, const float *t_ends, const float *background, float *weights,
                             float *trans, float *comp_rgb, float *opacity, float *depth, uint32_t n_rays, void *stream);
int nsr_bg_composite_backward(const int32_t *packed_info, const float *out16, float density_bias, const float *rgb_raw,
                              const float *weights, const float *trans, const float *t_starts, const float *t_ends,
                              const float *background, const float *d_comp_rgb, float *d_logit, float *d_rgb_raw,
                              uint32_t n_rays, void *stream);
/* d_out16[i] = [d_logit[i] + d_tex_in[i][0] | d_tex_in[i][1:n_feat] | 0] */
int nsr_bg_join_gradients(const float *d_logit, const float *d_tex_in, uint32_t stride, uint32_t n_feat, float *d_out16,
                          uint32_t n, const int32_t *n_dev, void *stream);

/* ---- masked mean losses of the systems' training steps (reference systems/nerf.py:97 `F.smooth_l1_loss(out['comp_rgb'][
 * out['rays_valid'][...,0]], batch['rgb'][out['rays_valid'][...,0]])`, systems/neus.py:98,102 `F.mse_loss` / `F.l1_loss` over
 * `rays_valid_full`): mean over the rows with mask != 0 WITHOUT the boolean-mask gathers (torch: a nonzero + a host
 * synchronisation each).  pred / target [n_rows][channels] fp32, mask [n_rows] bytes (a torch.bool tensor).
 * kind 0 smooth-L1 (beta), 1 MSE, 2 L1, 3 Huber (delta = beta): torch.nn.functional semantics, reduction "mean".
 * out [nsr_masked_loss_out_floats()]: out[0] = loss (0 when no row is valid -- torch returns NaN there), out[1] = number of
 * selected elements, the rest = per-workgroup partial sums, added by one wave in index order: bit-reproducible.
 * backward: d_pred = *grad_out * d loss / d pred (0 in masked-out rows); forward_out = the forward's `out`. ---- */
uint32_t nsr_masked_loss_out_floats(void);
int nsr_masked_loss_forward(const float *pred, const float *target, const uint8_t *mask, uint32_t n_rows, uint32_t channels,
                            int kind, float beta, float *out, void *stream);
int nsr_masked_loss_backward(const float *pred, const float *target, const uint8_t *mask, uint32_t n_rows, uint32_t channels,
                             int kind, float beta, const float *forward_out, const float *grad_out, float *d_pred,
                             void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NSR_HIP_H */
