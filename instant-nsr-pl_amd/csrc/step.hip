// Native orchestration of the fused NeRF training step (host code only: no kernels here).
//
// Measured on MI355X (profiles/r01_c_timeline_tail.csv): with every launch issued from Python the main queue idles ~40 %
// of the step -- ~10-25 us of interpreter + ctypes + allocator work per launch against kernels that take 3-20 us.
// So each PHASE of the step is one C call that carves its buffers out of one caller-provided workspace and issues
// all of its launches back to back on the caller's stream:
//
//   nsr_nerf_prune_pass : positions -> hash encode -> density MLP -> visibility prefix -> packed_info of the kept
//                         samples            (= the sigma_fn pass inside ray_marching, reference models/nerf.py:65-93)
//   nsr_nerf_main_pass  : kept rows copied (no re-encode) -> texture input -> colour MLP -> composite -> loss ->
//                         composite backward -> colour-MLP backward -> density-MLP backward -> table backward
//                                             (= models/nerf.py:95-109 + systems/nerf.py:97 + loss.backward())
//
// Two ways to pass the data-dependent sizes (M marched, S kept samples): as host integers -- the caller reads them back,
// two syncs per step -- or, with n_marched_dev / n_kept_dev, as DEVICE counts: then M and S are buffer capacities, every
// launch is made for the capacity and reads the live count itself, and a whole step is queued without a host sync
// (nsr/fused.py: forward_backward_async).  The caller sizes the workspaces with nsr_nerf_*_layout(); nothing is
// allocated here.  The item binning of the table backward runs on a helper stream owned by this file.
#include <string.h>

#include <vector>

#include "nsr_common.h"
#include <stdlib.h>

namespace {

inline uint64_t align_up(uint64_t v) { return (v + 255ull) & ~255ull; }

struct Carver {
    uint64_t off = 0;
    uint64_t take(uint64_t bytes)
    {
        const uint64_t o = off;
        off = align_up(off + bytes);
        return o;
    }
};

}  // namespace

// ---- opt-in HIP-event timing of the heavy launches (bench.py's roofline leg) --------------------------------------
// Events are recorded on the stream the kernels are launched on; disabled (zero cost) unless nsr_profile_enable(1).
namespace {
struct ProfRec { int tag; uint32_t units; hipEvent_t a, b; };
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
std::vector<hipEvent_t> g_prof_pool;  // events are reused: creating one costs the host several microseconds
inline hipEvent_t prof_event()
{
    if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

struct ProfScope {
    bool on;
    ProfRec r;
    hipStream_t st;
    ProfScope(int tag, uint32_t units, void *stream) : on(g_prof_on), st((hipStream_t)stream)
    {
        if (!on) return;
        r.tag = tag;
        r.units = units;
        r.a = prof_event();
        r.b = prof_event();
        (void)hipEventRecord(r.a, st);
    }
    ~ProfScope()
    {
        if (!on) return;
        (void)hipEventRecord(r.b, st);
        g_prof.push_back(r);
    }
};
}  // namespace

// ---- opt-in HOST timing of the main pass (NSR_HOST_TIMING=1): where the host's time to queue the pass goes, printed at exit
#include <chrono>
namespace {
struct HostTimer {
    bool on = getenv("NSR_HOST_TIMING") != nullptr;
    double acc[16] = {};
    uint64_t calls = 0;
    std::chrono::steady_clock::time_point t;
    void start() { if (on) { t = std::chrono::steady_clock::now(); ++calls; } }
    void mark(int i)
    {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        acc[i] += std::chrono::duration<double, std::micro>(now - t).count();
        t = now;
    }
    ~HostTimer()
    {
        if (!on || !calls) return;
        static const char *names[16] = {"set-up", "copy kept rows", "fork record + colour MLP forward", "composite forward",
                                        "helper: wait + binning + join record", "composite backward", "colour MLP backward",
                                        "density MLP backward", "join wait + table backward", "wgrad join", "", "", "", "", "", ""};
        fprintf(stderr, "nsr_nerf_main_pass host time over %llu calls (us per call):\n", (unsigned long long)calls);
        for (int i = 0; i < 10; ++i) fprintf(stderr, "  %-40s %7.2f\n", names[i], acc[i] / (double)calls);
    }
} g_ht;
}  // namespace

extern "C" void nsr_profile_enable(int on) { g_prof_on = on != 0; }

extern "C" int nsr_profile_collect(int tag, double *total_ms, uint64_t *launches, uint64_t *units)
{
    NSR_REQUIRE(total_ms && launches && units, "nsr_profile_collect: NULL pointer");
    *total_ms = 0.0;
    *launches = 0;
    *units = 0;
    if (tag < 0) {  // reset
        for (auto &r : g_prof) { g_prof_pool.push_back(r.a); g_prof_pool.push_back(r.b); }
        g_prof.clear();
        return NSR_OK;
    }
    for (auto &r : g_prof) {
        if (r.tag != tag) continue;
        if (hipEventSynchronize(r.b) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            *total_ms += ms;
            *launches += 1;
            *units += r.units;
        }
    }
    return NSR_OK;
}

#define NSR_TRY(expr)            \
    do {                         \
        const int rc_ = (expr);  \
        if (rc_ != NSR_OK) return rc_; \
    } while (0)

// Helper stream of the main pass: the item binning of the table backward depends only on the sample positions, so it runs
// beside the colour MLP / compositing / MLP backward chain instead of in front of the accumulation kernel.
// (ONE helper stream: HIP maps streams to a pool of 4 hardware queues in first-use order; every further stream measured slower
// -- profiles/r05_step_variants_*.json -- and is gone with the forms that used it)
struct HelperEvents { hipEvent_t fork = nullptr, join = nullptr, join_wgrad = nullptr, fork_wgrad = nullptr, dgrad_done = nullptr; };
struct HelperStream {
    hipStream_t stream = nullptr;
    HelperEvents ev;
    bool ok = false;
    bool init()
    {
        if (ok) return true;
        if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) return false;
        for (hipEvent_t *e : {&ev.fork, &ev.join, &ev.join_wgrad, &ev.fork_wgrad, &ev.dgrad_done})
            if (hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) return false;
        return ok = true;
    }
};
static HelperStream g_helper;  // one per process (= per GPU: one process per GPU)
static int g_variant[12] = {1, 0, 1, 0, 0, 1, 0, 0, 0, 128, 0, 0};  // nsr_nerf_step_variant (below)
#define HEV (g_helper.ev)

// the helper stream's handle (created on first use), for callers that queue follow-up work behind the weight-gradient
// kernels themselves (the trainer's optimizer launch for the MLP weights); NULL if it could not be created
extern "C" void *nsr_nerf_helper_stream(void) { return g_helper.init() ? (void *)g_helper.stream : nullptr; }

// `stream` waits for the point of the LAST main pass where the kept rows exist (the event the pass records for its own item
// binning): what a caller queues behind the pruning pass on another stream -- the next step's ray count and packing -- can
// hang on it instead of costing the main stream an event record of its own
extern "C" int nsr_nerf_wait_kept_rows(void *stream)
{
    NSR_REQUIRE(g_helper.ok, "nsr_nerf_wait_kept_rows: no main pass has run");
    NSR_REQUIRE(hipStreamWaitEvent((hipStream_t)stream, HEV.fork, 0) == hipSuccess, "nsr_nerf_wait_kept_rows: hipStreamWaitEvent failed");
    return NSR_OK;
}

// Forms of the step's kernels that stay switchable for the same-process A/B block of bench.py (`step_forms_ab`: this round's
// step against the round-4 forms of the same kernels) and as a fallback:
//   key 0: both networks' data gradients in ONE kernel (nsr_mlp_dgrad_pair) instead of two launches + a d_feature round trip
//   key 2: sample-partitioned compositing (nsr_composite_*_samples) instead of one wave per ray
//   key 5: the pass's fork events ride on the kernels in front of them (hipExtLaunchKernelGGL stop event) instead of being
//          recorded behind them
//   key 9: block cap of the pass's weight-gradient launches (nsr_mlp_wgrad_max_blocks, set around the pass's own launches only;
//          default 128 -- measured in the step: 512 -> 0.373, 256 -> 0.368, 128 -> 0.365, 64 -> 0.370, 32 -> 0.391 ms: with the
//          join deferred to the next density MLP fewer 64 KB-LDS blocks leave the table backward more of the chip; 0 = leave it)
// The forms rounds 4-5 measured and rejected -- dense levels through merged atomics, two weight-gradient streams, weight gradients
// behind the table backward, release-to-device events, a different host issue order, pipelined half encodes, the table backward
// on the helper stream, the ray-ordered sigma pass, ray-partitioned flat compositing -- are recorded in profiles/r04_*, r05_*,
// r06_step_variants_compositing.json and DESIGN.md section 5.1; their code is gone.
// value < 0 queries; returns the previous value.

extern "C" int nsr_nerf_step_variant(int key, int value)
{
    if (key < 0 || key >= 12) return -1;
    const int old = g_variant[key];
    if (value >= 0) g_variant[key] = value;
    return old;
}

// One-shot: the NEXT pruning pass makes its stream wait for `event` between its hash encode and its density MLP (the first
// kernel that reads network weights).  A trainer whose optimizer launch for the MLP weights runs on the helper stream hands
// that launch's event over instead of making the step's stream wait for it in front of the encode: the weight-gradient
// kernels + the optimizer then have the encode's duration to finish.  NULL clears.  The caller must make its stream wait for
// the event itself before anything ELSE reads the weights on it (occupancy refresh, evaluation, checkpoints).
// (keyed to its OWNER -- the step descriptor the trainer's pruning passes are called with: a pruning pass of any other
// FusedNeRFStep, trainer or evaluation in the process leaves it armed; owner == NULL: whoever comes next)
static struct { hipEvent_t event = nullptr; const void *owner = nullptr; } g_wait_before_mlp;
extern "C" int nsr_nerf_wait_before_mlp(const NsrNerfStepDesc *owner, void *event)
{
    if (!event && owner && g_wait_before_mlp.owner && g_wait_before_mlp.owner != owner) return NSR_OK;  // (not this caller's to clear)
    g_wait_before_mlp.event = (hipEvent_t)event;
    g_wait_before_mlp.owner = event ? (const void *)owner : nullptr;
    return NSR_OK;
}

// nsr_nerf_prune_pass_deferred leaves the packing of the kept counts to the next main pass's kept-row copy
struct DeferredPack {
    bool pending = false;
    const int32_t *kept = nullptr, *sums = nullptr;
    int32_t *packed = nullptr, *total = nullptr, *stats = nullptr;
    uint32_t n_rays = 0, capacity = 0;
};
static DeferredPack g_deferred;
constexpr uint64_t PRUNE_SUMS_BYTES = 65536 / 8 * 4;  // block sums of up to 65,536 rays, behind the prune workspace's rows

// 1: the main pass leaves the join with its weight-gradient kernels to the caller, who queues more work behind them on the
// helper stream (the optimizer launch) and makes the main stream wait for THAT instead -- one wait at the end of a step, not two
static bool g_defer_wgrad_join = false;
extern "C" int nsr_nerf_defer_wgrad_join(int on)
{
    const int old = g_defer_wgrad_join ? 1 : 0;
    g_defer_wgrad_join = on != 0;
    return old;
}

extern "C" int nsr_nerf_prune_layout(const NsrNerfStepDesc *d, uint32_t n_marched, NsrNerfPruneLayout *out)
{
    NSR_REQUIRE(d && out, "nsr_nerf_prune_layout: NULL pointer");
    const uint64_t M = n_marched, C = (uint64_t)d->grid.n_levels * d->grid.n_features;
    Carver c;
    out->x01 = c.take(M * 3 * 4);
    out->enc = c.take(M * C * 2);
    out->out1 = c.take(M * 16 * 2);
    out->acts1 = c.take(M * 64 * 2 * d->mlp_density.n_hidden);
    (void)c.take(PRUNE_SUMS_BYTES);  // (nsr_nerf_prune_pass_deferred: prune_sums_offset)
    out->total_bytes = c.off;
    return NSR_OK;
}

static uint64_t prune_sums_offset(const NsrNerfStepDesc *d, uint32_t n_marched)
{
    NsrNerfPruneLayout L;
    (void)nsr_nerf_prune_layout(d, n_marched, &L);
    return L.total_bytes - align_up(PRUNE_SUMS_BYTES);
}

static int prune_pass(const NsrNerfStepDesc *d, const float *rays_o, const float *rays_d,
                      const int64_t *ray_indices, const float *t_starts, const float *t_ends,
                      const int32_t *packed_info, const nsr_half *table, const nsr_half *w_density,
                      void *workspace, int32_t *kept_counts, int32_t *packed_kept, int32_t *total_kept,
                      uint32_t n_marched, uint32_t n_rays, const int32_t *n_marched_dev,
                      uint32_t kept_capacity, int32_t *kept_stats, const float *x01_marched, void *stream, bool defer_pack)
{
    NSR_REQUIRE(d && workspace && kept_counts && packed_kept && total_kept, "nsr_nerf_prune_pass: NULL pointer");
    g_deferred.pending = false;
    NsrNerfPruneLayout L;
    NSR_TRY(nsr_nerf_prune_layout(d, n_marched, &L));
    char *ws = (char *)workspace;
    // x01_marched: the caller already computed the unit-cube positions of the marched samples (the asynchronous trainer
    // does it on its marching stream, ahead of the step) -- otherwise they are formed here, into the workspace
    const float *x01 = x01_marched ? x01_marched : (const float *)(ws + L.x01);
    nsr_half *enc = (nsr_half *)(ws + L.enc), *out1 = (nsr_half *)(ws + L.out1), *acts1 = (nsr_half *)(ws + L.acts1);
    const uint32_t C = d->grid.n_levels * d->grid.n_features;
    if (!x01_marched)
        NSR_TRY(nsr_sample_positions_unit(rays_o, rays_d, ray_indices, t_starts, t_ends, d->radius, d->contraction,
                                          (float *)(ws + L.x01), nullptr, n_marched, n_marched_dev, stream));
    {
        {
            ProfScope p(NSR_PROF_GRID_FORWARD, n_marched, stream);
            NSR_TRY(nsr_hashgrid_forward_ex(x01, table, enc, n_marched, C, 1, d->grid.n_levels, &d->grid, n_marched_dev,
                                            stream));
        }
        if (g_wait_before_mlp.event && (!g_wait_before_mlp.owner || g_wait_before_mlp.owner == (const void *)d)) {
            // (nsr_nerf_wait_before_mlp: the network weights of this step are final behind this event)
            NSR_REQUIRE(hipStreamWaitEvent((hipStream_t)stream, g_wait_before_mlp.event, 0) == hipSuccess,
                        "nsr_nerf_prune_pass: hipStreamWaitEvent failed");
            g_wait_before_mlp.event = nullptr;
            g_wait_before_mlp.owner = nullptr;
        }
        {
            ProfScope p(NSR_PROF_MLP_FORWARD_DENSITY, n_marched, stream);
            NSR_TRY(nsr_mlp_forward_ex(enc, 0, C, d->grid.n_features, w_density, out1, acts1, n_marched, &d->mlp_density,
                                       n_marched_dev, stream));
        }
        // deferred packing: the kept-row copy of the main pass that follows forms the offsets itself from per-block sums
        // (the step's usual network shapes only: what nsr_nerf_copy_kept_rows covers)
        if (defer_pack && n_rays > 0 && n_rays <= 65536u && d->grid.n_features == 2 && d->mlp_density.n_hidden <= 2) {
            int32_t *sums = (int32_t *)(ws + prune_sums_offset(d, n_marched));
            NSR_TRY(nsr_visibility_prefix_sums(out1, 16, d->density_bias, t_starts, t_ends, packed_info, d->early_stop_eps,
                                               kept_counts, sums, n_rays, stream));
            g_deferred.pending = true;
            g_deferred.kept = kept_counts; g_deferred.sums = sums; g_deferred.packed = packed_kept;
            g_deferred.total = total_kept; g_deferred.stats = kept_stats; g_deferred.n_rays = n_rays;
            g_deferred.capacity = kept_capacity;
            return NSR_OK;
        }
        NSR_TRY(nsr_visibility_prefix(out1, 16, d->density_bias, t_starts, t_ends, packed_info, d->early_stop_eps,
                                      kept_counts, n_rays, stream));
    }
    NSR_TRY(nsr_pack_from_counts_capped(kept_counts, packed_kept, total_kept, n_rays, kept_capacity, kept_stats, nullptr,
                                        stream));
    return NSR_OK;
}

extern "C" int nsr_nerf_prune_pass(const NsrNerfStepDesc *d, const float *rays_o, const float *rays_d,
                                   const int64_t *ray_indices, const float *t_starts, const float *t_ends,
                                   const int32_t *packed_info, const nsr_half *table, const nsr_half *w_density,
                                   void *workspace, int32_t *kept_counts, int32_t *packed_kept, int32_t *total_kept,
                                   uint32_t n_marched, uint32_t n_rays, const int32_t *n_marched_dev,
                                   uint32_t kept_capacity, int32_t *kept_stats, const float *x01_marched, void *stream)
{
    return prune_pass(d, rays_o, rays_d, ray_indices, t_starts, t_ends, packed_info, table, w_density, workspace, kept_counts,
                      packed_kept, total_kept, n_marched, n_rays, n_marched_dev, kept_capacity, kept_stats, x01_marched,
                      stream, false);
}

// The same pass for a caller that queues nsr_nerf_main_pass* / nsr_nerf_render_forward on the SAME stream with the SAME
// packed_kept / total_kept right behind it: packed_kept, total_kept and the statistics are then written by that pass's first
// kernel (the kept-row copy forms the offsets itself) -- the one-workgroup scan between the two is gone from the step's chain.
// Nothing else may read packed_kept / total_kept in between.  Falls back to the plain pass for shapes the copy does not cover.
extern "C" int nsr_nerf_prune_pass_deferred(const NsrNerfStepDesc *d, const float *rays_o, const float *rays_d,
                                            const int64_t *ray_indices, const float *t_starts, const float *t_ends,
                                            const int32_t *packed_info, const nsr_half *table, const nsr_half *w_density,
                                            void *workspace, int32_t *kept_counts, int32_t *packed_kept,
                                            int32_t *total_kept, uint32_t n_marched, uint32_t n_rays,
                                            const int32_t *n_marched_dev, uint32_t kept_capacity, int32_t *kept_stats,
                                            const float *x01_marched, void *stream)
{
    return prune_pass(d, rays_o, rays_d, ray_indices, t_starts, t_ends, packed_info, table, w_density, workspace, kept_counts,
                      packed_kept, total_kept, n_marched, n_rays, n_marched_dev, kept_capacity, kept_stats, x01_marched,
                      stream, true);
}

extern "C" int nsr_nerf_main_layout(const NsrNerfStepDesc *d, uint32_t n_kept, uint32_t n_rays, NsrNerfMainLayout *out)
{
    NSR_REQUIRE(d && out, "nsr_nerf_main_layout: NULL pointer");
    const uint64_t S = n_kept, R = n_rays, C = (uint64_t)d->grid.n_levels * d->grid.n_features;
    Carver c;
    out->ray_indices = c.take(S * 8);
    out->t_starts = c.take(S * 4);
    out->t_ends = c.take(S * 4);
    out->weights = c.take(S * 4);
    out->comp_rgb = c.take(R * 3 * 4);
    out->opacity = c.take(R * 4);
    out->depth = c.take(R * 4);
    // (loss sum, valid rays) + the per-block partials the composite forward leaves for the composite backward
    out->loss_acc = c.take((2 + nsr_composite_l1_partials_floats(n_rays)) * 4);
    out->trans = c.take(S * 4);
    out->x01 = c.take(S * 3 * 4);
    out->dirs = c.take(S * 3 * 4);
    out->enc = c.take(S * C * 2);
    out->out1 = c.take(S * 16 * 2);
    out->acts1 = c.take(S * 64 * 2 * d->mlp_density.n_hidden);
    out->tex_in = c.take(S * 32 * 2);
    out->out2 = c.take(S * 16 * 2);
    out->acts2 = c.take(S * 64 * 2 * d->mlp_color.n_hidden);
    out->g_comp = c.take(R * 3 * 4);
    out->d_rgb = c.take(S * 3 * 4);
    out->d_logit = c.take(S * 4);
    out->d_tex = c.take(S * 32 * 4);
    out->d_enc = c.take(S * C * 4);
    out->partials = c.take(4 * (nsr_mlp_backward_workspace_floats(&d->mlp_color, n_kept) +
                                nsr_mlp_backward_workspace_floats(&d->mlp_density, n_kept)));
    out->grid_ws = c.take(4 * nsr_hashgrid_backward_params_workspace_floats(&d->grid, n_kept));
    out->total_bytes = c.off;
    return NSR_OK;
}

static int main_pass(const NsrNerfStepDesc *d, const void *prune_workspace, uint32_t n_marched,
                     const int32_t *packed_marched, const int32_t *packed_kept, const float *t_starts,
                     const float *t_ends, const float *rays_d, const float *background, const float *gt_rgb,
                     const nsr_half *w_density, const nsr_half *w_color, float *grad_density_mlp,
                     float *grad_table, float *grad_color_mlp, void *workspace, uint32_t n_kept,
                     uint32_t n_rays, int compute_grads, const int32_t *n_kept_dev,
                     const float *x01_marched, const NsrTableAdam *table_adam, const NsrTableExchange *xchg, void *stream,
                     int phases = 3 /* 1: forward (+ item binning when compute_grads), 2: backward, 3: both */,
                     const NsrRenderGrads *up = nullptr /* upstream gradients; NULL: the masked smooth-L1 loss on gt_rgb */)
{
    NSR_REQUIRE(d && prune_workspace && workspace && packed_marched && packed_kept, "nsr_nerf_main_pass: NULL pointer");
    NSR_REQUIRE(d->mlp_color.n_in == 32 && d->mlp_density.n_out == 16, "nsr_nerf_main_pass: the texture input is "
                "[16 features | 16 SH] (reference models/texture.py:26 with feature_dim 16)");
    g_ht.start();
    NsrNerfPruneLayout P;
    NsrNerfMainLayout L;
    NSR_TRY(nsr_nerf_prune_layout(d, n_marched, &P));
    NSR_TRY(nsr_nerf_main_layout(d, n_kept, n_rays, &L));
    const char *pw = (const char *)prune_workspace;
    char *ws = (char *)workspace;
    const uint32_t C = d->grid.n_levels * d->grid.n_features, S = n_kept;
    hipStream_t st = (hipStream_t)stream;
    float *acc = (float *)(ws + L.loss_acc);
    nsr_half *enc = (nsr_half *)(ws + L.enc), *out1 = (nsr_half *)(ws + L.out1), *acts1 = (nsr_half *)(ws + L.acts1);
    nsr_half *tex_in = (nsr_half *)(ws + L.tex_in), *out2 = (nsr_half *)(ws + L.out2), *acts2 = (nsr_half *)(ws + L.acts2);
    float *x01 = (float *)(ws + L.x01);
    float *t0 = (float *)(ws + L.t_starts), *t1 = (float *)(ws + L.t_ends);
    float *weights = (float *)(ws + L.weights), *trans = (float *)(ws + L.trans);
    float *comp_rgb = (float *)(ws + L.comp_rgb), *opacity = (float *)(ws + L.opacity), *depth = (float *)(ws + L.depth);

    // kept rows of everything the sigma pass computed (a per-ray memcpy; nothing is re-encoded)
    const uint32_t nh1 = d->mlp_density.n_hidden;
    const float *x01m = x01_marched ? x01_marched : (const float *)(pw + P.x01);
    const void *src[8] = {t_starts, t_ends, x01m, pw + P.enc, pw + P.out1, pw + P.acts1, nullptr, nullptr};
    void *dst[8] = {t0, t1, x01, enc, out1, acts1, nullptr, nullptr};
    // the encoding is level-major [L][n][F] (fp16): L planes of F*2-byte rows
    const uint32_t F = d->grid.n_features, Lv = d->grid.n_levels;
    uint32_t rb[8] = {4, 4, 12, F * 2, 32, 128, 0, 0};
    uint32_t planes[8] = {1, 1, 1, Lv, 1, 1, 1, 1};
    uint64_t sp[8] = {0, 0, 0, (uint64_t)n_marched * F * 2, 0, 0, 0, 0};
    uint64_t dp[8] = {0, 0, 0, (uint64_t)S * F * 2, 0, 0, 0, 0};
    uint32_t na = 6;
    for (uint32_t h = 1; h < nh1 && na < 8; ++h, ++na) {  // further hidden layers of the density MLP
        src[na] = pw + P.acts1 + (uint64_t)h * n_marched * 128;
        dst[na] = (char *)acts1 + (uint64_t)h * S * 128;
        rb[na] = 128;
    }
    NSR_REQUIRE(F * 2 % 4 == 0, "nsr_nerf_main_pass: n_features_per_level must be even");
    const bool overlap_bins = compute_grads && S > 0 && g_helper.init();
    // forward and backward in one call with the built-in loss: the (loss sum, valid rays) reduction is folded into the two
    // compositing kernels instead of a one-workgroup kernel between them (NSR_L1_SEPARATE: A/B switch)
    static const bool l1_separate = getenv("NSR_L1_SEPARATE") != nullptr;
    const bool l1_folded = phases == 3 && compute_grads && gt_rgb && !up && S > 0 && n_rays > 0 && !l1_separate;
    const bool flat = g_variant[2] != 0;  // (nsr_nerf_step_variant key 2: sample-partitioned compositing)
    // (a stream that is being captured into a graph takes plain launches + event records only)
    hipStreamCaptureStatus capture_status = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &capture_status);
    const bool capturing = capture_status != hipStreamCaptureStatusNone;
    g_ht.mark(0);
    if (phases & 1) {
    const bool deferred = g_deferred.pending;
    g_deferred.pending = false;
    if (deferred)
        NSR_REQUIRE(g_deferred.packed == packed_kept && g_deferred.n_rays == n_rays && (!n_kept_dev || g_deferred.total == n_kept_dev),
                    "nsr_nerf_main_pass: the pass behind nsr_nerf_prune_pass_deferred must take the same packed_kept / "
                    "total_kept / n_rays");
    if (deferred && S == 0)  // (no sample buffer: nothing to copy, the packing is still owed)
        NSR_TRY(nsr_pack_from_counts_capped(g_deferred.kept, g_deferred.packed, g_deferred.total, n_rays, g_deferred.capacity,
                                            g_deferred.stats, nullptr, stream));
    // key 5: the fork events of the pass ride on the kernels they follow (NSR_LAUNCH_STOP) instead of being recorded behind them
    bool fork_pending = true;
    if (overlap_bins && g_variant[5] && !capturing && S > 0 && F == 2 && nh1 <= 2) {
        nsr_next_stop_event = HEV.fork;
        fork_pending = false;
    }
    if (S > 0) {  // nothing kept (e.g. an empty occupancy grid): the per-ray outputs below are still produced
        if (deferred)  // packing folded into the copy: packed_kept / total_kept are written by this launch
            NSR_TRY(nsr_nerf_copy_kept_rows_scan(packed_marched, g_deferred.kept, g_deferred.sums, g_deferred.packed,
                                                 g_deferred.total, g_deferred.stats, t_starts, t_ends, x01m,
                                                 (const nsr_half *)(pw + P.enc), (const nsr_half *)(pw + P.out1),
                                                 (const nsr_half *)(pw + P.acts1), t0, t1, x01, enc, out1, acts1, Lv, nh1,
                                                 n_marched, S, rays_d, (int64_t *)(ws + L.ray_indices), tex_in, n_rays,
                                                 stream));
        else if (F == 2 && nh1 <= 2)  // the step's usual shape: one lane per kept sample, all its rows in one round trip
            NSR_TRY(nsr_nerf_copy_kept_rows(packed_marched, packed_kept, t_starts, t_ends, x01m,
                                            (const nsr_half *)(pw + P.enc), (const nsr_half *)(pw + P.out1),
                                            (const nsr_half *)(pw + P.acts1), t0, t1, x01, enc, out1, acts1, Lv, nh1,
                                            n_marched, S, rays_d, (int64_t *)(ws + L.ray_indices), tex_in, n_rays,
                                            stream));
        else
            NSR_TRY(nsr_copy_ray_prefix_rows_ex(packed_marched, packed_kept, na, src, dst, rb, planes, sp, dp, rays_d,
                                                nullptr, (int64_t *)(ws + L.ray_indices),
                                                (const nsr_half *)(pw + P.out1), 16, tex_in, n_rays, stream));
    }
    // fork: the table-backward items are binned on the helper stream as soon as the kept positions exist.  The EVENT is
    // recorded here; the helper's launches are queued BEHIND the colour MLP and the compositing forward -- the host needs
    // ~7 us per launch, and with the four binning launches queued first the main stream sat idle for 30-45 us waiting for
    // its next kernel (rocprofv3 timeline: copy_kept_rows ... 46 us ... mlp_forward)
    g_ht.mark(1);
    if (overlap_bins && nsr_next_stop_event == HEV.fork) {  // the copy did not take it (the generic row copy)
        nsr_next_stop_event = nullptr;
        fork_pending = true;
    }
    if (overlap_bins && fork_pending)
        NSR_REQUIRE(hipEventRecord(HEV.fork, st) == hipSuccess, "nsr_nerf_main_pass: helper stream fork failed");
    {
        ProfScope p(NSR_PROF_MLP_FORWARD_COLOR, S, stream);
        NSR_TRY(nsr_mlp_forward_ex(tex_in, 0, 32, 0, w_color, out2, compute_grads ? acts2 : nullptr, S, &d->mlp_color,
                                   n_kept_dev, stream));
    }
    g_ht.mark(2);
    if (flat) {  // one lane per sample, a wave per 64 samples (csrc/fused.hip k_composite_forward_samples); the loss
                        // partials ride along when folded
        NSR_TRY(nsr_composite_forward_samples(out1, 16, d->density_bias, t0, t1, out2, 16, packed_kept,
                                              (const int64_t *)(ws + L.ray_indices), background, weights, trans, comp_rgb,
                                              opacity, depth, l1_folded ? gt_rgb : nullptr, l1_folded ? acc + 2 : nullptr,
                                              n_rays, S, n_kept_dev, stream));
        if (!l1_folded && gt_rgb)
            NSR_TRY(nsr_smooth_l1_valid_set(comp_rgb, opacity, gt_rgb, acc, n_rays, stream));
    } else if (l1_folded) {  // the loss reduction rides in the two compositing kernels (per-block partials behind acc)
        NSR_TRY(nsr_composite_forward_smooth_l1(out1, 16, d->density_bias, t0, t1, out2, 16, packed_kept, background, weights,
                                                trans, comp_rgb, opacity, depth, gt_rgb, acc + 2, n_rays, stream));
    } else {
        NSR_TRY(nsr_composite_forward(out1, 16, d->density_bias, t0, t1, out2, 16, packed_kept, background, weights, trans,
                                      comp_rgb, opacity, depth, n_rays, stream));
        if (gt_rgb)
            NSR_TRY(nsr_smooth_l1_valid_set(comp_rgb, opacity, gt_rgb, acc, n_rays, stream));  // writes acc: no memset needed
    }
    g_ht.mark(3);
    if (overlap_bins) {
        NSR_REQUIRE(hipStreamWaitEvent(g_helper.stream, HEV.fork, 0) == hipSuccess,
                    "nsr_nerf_main_pass: helper stream fork failed");
        {
            ProfScope p(NSR_PROF_GRID_BACKWARD_BIN, S, g_helper.stream);
            NSR_TRY(nsr_hashgrid_backward_params_owner_bin(x01, (float *)(ws + L.grid_ws), S, d->grid.n_levels, &d->grid,
                                                           n_kept_dev, g_helper.stream));
        }
        NSR_REQUIRE(hipEventRecord(HEV.join, g_helper.stream) == hipSuccess,
                    "nsr_nerf_main_pass: helper stream join failed");
    }
    }  // phases & 1
    g_ht.mark(4);
    if (!(phases & 2)) return NSR_OK;
    NSR_REQUIRE(!compute_grads || up || gt_rgb, "nsr_nerf_main_pass: no loss (gt_rgb) and no upstream gradients");
    NSR_REQUIRE(!(table_adam && compute_grads && S == 0), "nsr_nerf_main_pass: the fused table update needs a non-empty "
                "sample buffer (the table still decays when nothing was kept)");
    if (xchg) {
        NSR_REQUIRE(compute_grads && !grad_table && !table_adam && xchg->grad_bf16 && xchg->n_groups >= 1 &&
                        xchg->n_groups <= 4 &&
                        xchg->grad_bf16_elems >= (uint64_t)d->grid.n_entries * d->grid.n_features,
                    "nsr_nerf_main_pass_exchange: needs a bf16 gradient buffer for the whole table, 1..4 level groups, and "
                    "neither grad_table nor table_adam");
        if (S == 0) {  // nothing kept on this rank: it still contributes (zeros) to every collective
            NSR_REQUIRE(hipMemsetAsync(xchg->grad_bf16, 0, xchg->grad_bf16_elems * 2, st) == hipSuccess,
                        "nsr_nerf_main_pass_exchange: hipMemsetAsync failed");
            if (xchg->event_small)
                NSR_REQUIRE(hipEventRecord((hipEvent_t)xchg->event_small, st) == hipSuccess, "hipEventRecord failed");
            for (uint32_t g = 0; g < xchg->n_groups; ++g)
                if (xchg->event_group[g])
                    NSR_REQUIRE(hipEventRecord((hipEvent_t)xchg->event_group[g], st) == hipSuccess, "hipEventRecord failed");
            return NSR_OK;
        }
    }
    if (compute_grads && S == 0 && grad_table)
        // nothing kept: the table gradient of this step is zero -- WRITTEN as zeros, because callers rely on the table backward
        // overwriting every entry (nsr/parallel.py ShardedAdamW.step(overwritten=...) never clears it: a stale gradient would
        // be sent into the reduce-scatter again)
        NSR_REQUIRE(hipMemsetAsync(grad_table, 0, (uint64_t)d->grid.n_entries * d->grid.n_features * sizeof(float), st) ==
                        hipSuccess, "nsr_nerf_main_pass: hipMemsetAsync failed");
    if (!compute_grads || S == 0) return NSR_OK;
    NSR_REQUIRE(grad_density_mlp && (grad_table || table_adam || xchg) && grad_color_mlp, "nsr_nerf_main_pass: NULL gradient buffer");
    float *d_rgb = (float *)(ws + L.d_rgb), *d_logit = (float *)(ws + L.d_logit);
    float *d_tex = (float *)(ws + L.d_tex), *d_enc = (float *)(ws + L.d_enc);
    float *part2 = (float *)(ws + L.partials);
    float *part1 = part2 + nsr_mlp_backward_workspace_floats(&d->mlp_color, S);
    // the weight-gradient kernels of both MLPs (8 short launches) run on the helper stream underneath the density MLP's
    // dgrad and the table backward -- only dx continues down the main chain
    static const bool wgrad_inline = getenv("NSR_WGRAD_INLINE") != nullptr;  // diagnostic A/B switch
    void *wg = (overlap_bins && !wgrad_inline) ? (void *)g_helper.stream : nullptr;
    const int64_t *ray_kept = (const int64_t *)(ws + L.ray_indices);
    if (flat && up)
        NSR_TRY(nsr_composite_backward_samples(out1, 16, d->density_bias, t0, t1, out2, 16, packed_kept, ray_kept, background,
                                               weights, trans, up->comp_rgb, up->opacity, up->depth, up->weights, nullptr,
                                               nullptr, nullptr, nullptr, nullptr, d->loss_scale, d_rgb, d_logit, n_rays, S,
                                               n_kept_dev, stream));
    else if (flat)  // built-in masked smooth-L1: (sum, valid rays) from the forward's partials when folded, from acc otherwise
        NSR_TRY(nsr_composite_backward_samples(out1, 16, d->density_bias, t0, t1, out2, 16, packed_kept, ray_kept, background,
                                               weights, trans, nullptr, nullptr, nullptr, nullptr, comp_rgb, opacity, gt_rgb,
                                               l1_folded ? acc + 2 : nullptr, acc, d->loss_scale, d_rgb, d_logit, n_rays, S,
                                               n_kept_dev, stream));
    else if (up)  // the caller's loss: arbitrary dL/d comp_rgb, dL/d opacity, dL/d depth (+ dL/d weights: distortion loss)
        NSR_TRY(nsr_composite_backward_ex(out1, 16, d->density_bias, t0, t1, out2, 16, packed_kept, background, weights, trans,
                                          up->comp_rgb, up->opacity, up->depth, up->weights, d_rgb, d_logit, n_rays, stream));
    else if (l1_folded)
        NSR_TRY(nsr_composite_backward_smooth_l1_partials(out1, 16, d->density_bias, t0, t1, out2, 16, packed_kept,
                                                          background, weights, trans, comp_rgb, opacity, gt_rgb, acc + 2, acc,
                                                          d->loss_scale, d_rgb, d_logit, n_rays, stream));
    else
        NSR_TRY(nsr_composite_backward_smooth_l1(out1, 16, d->density_bias, t0, t1, out2, 16, packed_kept, background,
                                                 weights, trans, comp_rgb, opacity, gt_rgb, acc, d->loss_scale, d_rgb,
                                                 d_logit, n_rays, stream));
    g_ht.mark(5);
    // (the weight-gradient block cap is a library-wide knob other callers -- the NeuS steps, the drop-in tcnn modules -- leave at
    // its default: lowered for this pass's launches only, restored on every way out)
    struct WgradCap {
        uint32_t old = 0;
        bool set = false;
        explicit WgradCap(int cap) { if (cap > 0) { old = nsr_mlp_wgrad_max_blocks((uint32_t)cap); set = true; } }
        ~WgradCap() { if (set) (void)nsr_mlp_wgrad_max_blocks(old); }
    } wgrad_cap(wg && g_defer_wgrad_join ? g_variant[9] : 0);
    const bool pair = g_variant[0] && nsr_mlp_dgrad_pair_supported(&d->mlp_color, &d->mlp_density) && C == 32;
    // the weight-gradient kernels + reductions of both networks (behind nsr_mlp_dgrad_pair), forked from `st` through `fork`
    auto queue_wgrads = [&](hipEvent_t fork, bool already_recorded) -> int {
        if (wg)
            NSR_REQUIRE((already_recorded || hipEventRecord(fork, st) == hipSuccess) &&
                            hipStreamWaitEvent(g_helper.stream, fork, 0) == hipSuccess,
                        "nsr_nerf_main_pass: weight-gradient fork failed");
        NSR_TRY(nsr_mlp_backward_phases(d_rgb, 1, 3, nullptr, out2, tex_in, 0, 32, 0, acts2, w_color, grad_color_mlp, nullptr, 32, 0,
                                        part2, S, d->grad_scale, &d->mlp_color, n_kept_dev, wg ? wg : stream, 2));
        NSR_TRY(nsr_mlp_backward_phases(d_enc, 1, 32, d_logit, out1, enc, 0, C, d->grid.n_features, acts1, w_density,
                                        grad_density_mlp, nullptr, C, d->grid.n_features, part1, S, d->grad_scale,
                                        &d->mlp_density, n_kept_dev, wg ? wg : stream, 2));
        return NSR_OK;
    };
    if (pair) {
        const bool ride = wg && g_variant[5] && !capturing && !g_prof_on;  // (the profiling scope records its own events)
        if (ride) nsr_next_stop_event = HEV.dgrad_done;
        {   // both networks' data gradients in one launch; d_feature stays in registers (csrc/mlp.hip k_mlp_dgrad_pair)
            ProfScope p(NSR_PROF_MLP_BACKWARD_COLOR, S, stream);
            NSR_TRY(nsr_mlp_dgrad_pair(d_rgb, d_logit, out2, acts2, w_color, part2, acts1, w_density, part1, d_enc, S,
                                       d->grad_scale, &d->mlp_color, &d->mlp_density, n_kept_dev, stream));
        }
        const bool rode = ride && nsr_next_stop_event == nullptr;
        nsr_next_stop_event = nullptr;
        NSR_TRY(queue_wgrads(HEV.dgrad_done, rode));
    } else {
    {
        ProfScope p(NSR_PROF_MLP_BACKWARD_COLOR, S, stream);
        NSR_TRY(nsr_mlp_backward_split(d_rgb, 1, 3, nullptr, out2, tex_in, 0, 32, 0, acts2, w_color, grad_color_mlp, d_tex,
                                       32, 0, part2, S, d->grad_scale, &d->mlp_color, n_kept_dev, stream, wg));
    }
    g_ht.mark(6);
    {
        ProfScope p(NSR_PROF_MLP_BACKWARD_DENSITY, S, stream);
        NSR_TRY(nsr_mlp_backward_split(d_tex, 1, 32, d_logit, out1, enc, 0, C, d->grid.n_features, acts1, w_density,
                                       grad_density_mlp, d_enc, C, d->grid.n_features, part1, S, d->grad_scale,
                                       &d->mlp_density, n_kept_dev, stream, wg));
    }
    }
    g_ht.mark(7);
    if (xchg && xchg->event_small)  // the MLP weight gradients are final behind what is queued on their stream by now
        NSR_REQUIRE(hipEventRecord((hipEvent_t)xchg->event_small, wg ? g_helper.stream : st) == hipSuccess,
                    "nsr_nerf_main_pass_exchange: hipEventRecord failed");
    if (xchg) {
        NSR_REQUIRE(overlap_bins, "nsr_nerf_main_pass_exchange: the helper stream is not available");
        NSR_REQUIRE(hipStreamWaitEvent(st, HEV.join, 0) == hipSuccess, "nsr_nerf_main_pass: helper stream join failed");
        ProfScope p(NSR_PROF_GRID_BACKWARD, S, stream);  // (all groups: one operation)
        for (uint32_t g = 0; g < xchg->n_groups; ++g) {
            NSR_TRY(nsr_hashgrid_backward_params_owner_accumulate_range(x01, d_enc, nullptr, xchg->grad_bf16,
                                                                        (float *)(ws + L.grid_ws), S, d->grid.n_levels,
                                                                        1.0f, xchg->level_begin[g], xchg->level_end[g],
                                                                        &d->grid, n_kept_dev, stream));
            if (xchg->event_group[g])
                NSR_REQUIRE(hipEventRecord((hipEvent_t)xchg->event_group[g], st) == hipSuccess,
                            "nsr_nerf_main_pass_exchange: hipEventRecord failed");
        }
    } else {
        ProfScope p(NSR_PROF_GRID_BACKWARD, S, stream);
        if (overlap_bins) {
            NSR_REQUIRE(hipStreamWaitEvent(st, HEV.join, 0) == hipSuccess,
                        "nsr_nerf_main_pass: helper stream join failed");
            if (table_adam)  // the optimizer's update of the table happens inside the backward (no gradient store)
                // (round 4: launching the small dense levels -- the slowest workgroups on a trained scene -- on a stream of their own
                // beside the other levels was built and measured: 125-135 us for that launch alone, step 0.511 -> 0.546 ms; their
                // chains are hidden better INSIDE the one launch, where they are dispatched first)
                NSR_TRY(nsr_hashgrid_backward_params_owner_accumulate_adam(x01, d_enc, 2, 0, (float *)(ws + L.grid_ws), S,
                                                                           d->grid.n_levels, 1.0f, &d->grid, n_kept_dev,
                                                                           table_adam, stream));
            else
                NSR_TRY(nsr_hashgrid_backward_params_owner_accumulate(x01, d_enc, 2, 0, grad_table,
                                                                      (float *)(ws + L.grid_ws), S, d->grid.n_levels,
                                                                      1.0f, 0, &d->grid, n_kept_dev, stream));
        } else {
            NSR_REQUIRE(!table_adam, "nsr_nerf_main_pass: the fused table update needs the helper stream");
            NSR_TRY(nsr_hashgrid_backward_params_owner(x01, d_enc, 2, 0, grad_table, (float *)(ws + L.grid_ws), S,
                                                       d->grid.n_levels, 1.0f, 0, &d->grid, n_kept_dev, stream));
        }
    }
    g_ht.mark(8);
    if (wg && !g_defer_wgrad_join)  // join: the optimizer step that follows on `stream` reads the MLP gradients
        NSR_REQUIRE(hipEventRecord(HEV.join_wgrad, g_helper.stream) == hipSuccess &&
                        hipStreamWaitEvent(st, HEV.join_wgrad, 0) == hipSuccess,
                    "nsr_nerf_main_pass: weight-gradient join failed");
    return NSR_OK;
}

extern "C" int nsr_nerf_main_pass(const NsrNerfStepDesc *d, const void *prune_workspace, uint32_t n_marched,
                                  const int32_t *packed_marched, const int32_t *packed_kept, const float *t_starts,
                                  const float *t_ends, const float *rays_d, const float *background, const float *gt_rgb,
                                  const nsr_half *w_density, const nsr_half *w_color, float *grad_density_mlp,
                                  float *grad_table, float *grad_color_mlp, void *workspace, uint32_t n_kept,
                                  uint32_t n_rays, int compute_grads, const int32_t *n_kept_dev,
                                  const float *x01_marched, const NsrTableAdam *table_adam, void *stream)
{
    return main_pass(d, prune_workspace, n_marched, packed_marched, packed_kept, t_starts, t_ends, rays_d, background,
                     gt_rgb, w_density, w_color, grad_density_mlp, grad_table, grad_color_mlp, workspace, n_kept, n_rays,
                     compute_grads, n_kept_dev, x01_marched, table_adam, nullptr, stream);
}

// ---- the same pass split at the loss, for callers that own it (the reference's systems/nerf.py:87-99 computes the loss in
// torch on the model's output dict and calls backward()): forward keeps every activation in `workspace` and -- when
// prepare_backward -- bins the table-backward items on the helper stream; backward takes the upstream gradients.
extern "C" int nsr_nerf_render_forward(const NsrNerfStepDesc *d, const void *prune_workspace, uint32_t n_marched,
                                       const int32_t *packed_marched, const int32_t *packed_kept, const float *t_starts,
                                       const float *t_ends, const float *rays_d, const float *background,
                                       const nsr_half *w_density, const nsr_half *w_color, void *workspace,
                                       uint32_t n_kept, uint32_t n_rays, int prepare_backward, const int32_t *n_kept_dev,
                                       const float *x01_marched, void *stream)
{
    return main_pass(d, prune_workspace, n_marched, packed_marched, packed_kept, t_starts, t_ends, rays_d, background,
                     nullptr, w_density, w_color, nullptr, nullptr, nullptr, workspace, n_kept, n_rays, prepare_backward,
                     n_kept_dev, x01_marched, nullptr, nullptr, stream, 1, nullptr);
}

extern "C" int nsr_nerf_render_backward(const NsrNerfStepDesc *d, const void *prune_workspace, uint32_t n_marched,
                                        const int32_t *packed_marched, const int32_t *packed_kept, const float *rays_d,
                                        const float *background, const NsrRenderGrads *upstream, const nsr_half *w_density,
                                        const nsr_half *w_color, float *grad_density_mlp, float *grad_table,
                                        float *grad_color_mlp, void *workspace, uint32_t n_kept, uint32_t n_rays,
                                        const int32_t *n_kept_dev, void *stream)
{
    NSR_REQUIRE(upstream && upstream->comp_rgb, "nsr_nerf_render_backward: upstream gradient of comp_rgb is NULL");
    return main_pass(d, prune_workspace, n_marched, packed_marched, packed_kept, nullptr, nullptr, rays_d, background,
                     nullptr, w_density, w_color, grad_density_mlp, grad_table, grad_color_mlp, workspace, n_kept, n_rays, 1,
                     n_kept_dev, nullptr, nullptr, nullptr, stream, 2, upstream);
}

extern "C" int nsr_nerf_main_pass_exchange(const NsrNerfStepDesc *d, const void *prune_workspace, uint32_t n_marched,
                                           const int32_t *packed_marched, const int32_t *packed_kept,
                                           const float *t_starts, const float *t_ends, const float *rays_d,
                                           const float *background, const float *gt_rgb, const nsr_half *w_density,
                                           const nsr_half *w_color, float *grad_density_mlp, float *grad_color_mlp,
                                           void *workspace, uint32_t n_kept, uint32_t n_rays, const int32_t *n_kept_dev,
                                           const float *x01_marched, const NsrTableExchange *exchange, void *stream)
{
    NSR_REQUIRE(exchange, "nsr_nerf_main_pass_exchange: exchange is NULL");
    return main_pass(d, prune_workspace, n_marched, packed_marched, packed_kept, t_starts, t_ends, rays_d, background,
                     gt_rgb, w_density, w_color, grad_density_mlp, nullptr, grad_color_mlp, workspace, n_kept, n_rays, 1,
                     n_kept_dev, x01_marched, nullptr, exchange, stream);
}
