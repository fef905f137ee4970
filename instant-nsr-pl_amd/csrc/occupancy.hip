// Device-side refresh of a nerfacc-style occupancy grid (OccupancyGrid._update of nerfacc 0.3.3, reached from the
// reference at models/nerf.py:45-55 every 16th step): cell selection, EMA update and re-binarisation WITHOUT the host
// ever seeing a count.  The torch formulation in nerfacc/grid.py needs torch.nonzero (a host sync that drains the queue of
// the asynchronous training step) and ~40 small kernels; here the selected-cell count stays on the device and the density
// evaluation in between runs through the (n, n_dev) entry points of the hash grid and the MLP.
//
//   step <  warmup: every cell.
//   step >= warmup: n_uniform cells drawn uniformly + the occupied cells (or n_uniform of them drawn with replacement when
//                   more than n_uniform are occupied)             -- nerfacc's _sample_uniform_and_occupied_cells
//   occs[c] <- max(occs_old[c] * decay, occ(c))   (gather-then-scatter: every update reads the OLD value, as torch does)
//   binary  <- occs > min(mean(occs), occ_thre)
#include "nsr_common.h"

namespace {

constexpr int EW_BLOCK = 256;

// occupied cells from the 4x4x4-brick bitfield, in brick order: popcount per brick + exclusive scan by ONE workgroup
__global__ void __launch_bounds__(1024)
k_occ_brick_scan(const unsigned long long *__restrict__ bricks, uint32_t n_bricks, uint32_t *__restrict__ brick_offset,
                 int32_t *__restrict__ n_occupied)
{
    __shared__ uint32_t wave_tot[16];
    const uint32_t tid = threadIdx.x, chunk = (n_bricks + 1023) / 1024;
    const uint32_t lo = min(tid * chunk, n_bricks), hi = min(lo + chunk, n_bricks);
    uint32_t s = 0;
    for (uint32_t b = lo; b < hi; ++b) s += (uint32_t)__popcll(bricks[b]);
    uint32_t v = s;
    const int lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    if (lane == 63) wave_tot[w] = v;
    __syncthreads();
    uint32_t prefix = 0;
    for (int k = 0; k < w; ++k) prefix += wave_tot[k];
    uint32_t run = prefix + v - s;
    for (uint32_t b = lo; b < hi; ++b) {
        brick_offset[b] = run;
        run += (uint32_t)__popcll(bricks[b]);
    }
    if (tid == 1023) *n_occupied = (int32_t)(prefix + v);
}

__global__ void __launch_bounds__(EW_BLOCK)
k_occ_brick_expand(const unsigned long long *__restrict__ bricks, const uint32_t *__restrict__ brick_offset, int3 res,
                   uint32_t n_bricks, uint32_t *__restrict__ occupied_cells)
{
    const uint32_t b = blockIdx.x * EW_BLOCK + threadIdx.x;
    if (b >= n_bricks) return;
    unsigned long long bits = bricks[b];
    if (!bits) return;
    const int nby = res.y >> 2, nbz = res.z >> 2;
    const int bz = b % nbz, by = (b / nbz) % nby, bx = b / (nbz * nby);
    uint32_t o = brick_offset[b];
    while (bits) {
        const int bit = __builtin_ctzll(bits);
        bits &= bits - 1ull;
        const int i = bit >> 4, j = (bit >> 2) & 3, k = bit & 3;  // bit = (i * 4 + j) * 4 + k, as k_pack_bricks packs it
        occupied_cells[o++] = (uint32_t)(((bx * 4 + i) * res.y + (by * 4 + j)) * res.z + (bz * 4 + k));
    }
}

// the cells to evaluate and their jittered positions in grid-unit coordinates.  A thread takes OCC_SPT consecutive slots.
// sphere: a grid on the contracted space -- samples outside its unit sphere are dropped HERE, as nerfacc drops them before
// it evaluates anything (grid.py: `mask = (x - 0.5).norm(dim=1) < 0.5`): 48 % of a cube's cells, whose positions have no
// preimage (their encode was a scatter of cache-missing gathers: 4.8 ms for a 128^3 grid instead of 1 ms).  The survivors of
// a workgroup's 4,096 slots stay in slot order and are appended with ONE atomic on *n_cells (cleared by the host) -- one per
// wave cost 2 ms for 2 M slots: ~60 ns per atomic on one address.
constexpr int OCC_SPT = 16;
__global__ void __launch_bounds__(EW_BLOCK)
k_occ_make_samples(const uint32_t *__restrict__ occupied_cells, const int32_t *__restrict__ n_occupied,
                   const float *__restrict__ u_cell, const float *__restrict__ u_pick, const float *__restrict__ jitter,
                   int3 res, uint32_t n_uniform, int all_cells, int sphere, uint32_t capacity, uint32_t *__restrict__ cells,
                   float *__restrict__ x_unit, int32_t *__restrict__ n_cells)
{
    __shared__ uint32_t wave_tot[EW_BLOCK / 64];
    __shared__ uint32_t block_base;
    const uint32_t n_total_cells = (uint32_t)res.x * res.y * res.z;
    const uint32_t n_occ = all_cells ? 0u : (uint32_t)max(*n_occupied, 0);
    const uint32_t n_take = min(n_occ, n_uniform);
    const uint32_t total = all_cells ? n_total_cells : min(n_uniform + n_take, capacity);
    const uint32_t i0 = (blockIdx.x * EW_BLOCK + threadIdx.x) * OCC_SPT;
    if (!sphere && i0 == 0) *n_cells = (int32_t)total;
    uint32_t c[OCC_SPT], keep = 0, n_keep = 0;
    float xu[OCC_SPT][3];
#pragma unroll
    for (int u = 0; u < OCC_SPT; ++u) {
        const uint32_t i = i0 + u;
        if (i >= total) { c[u] = 0; continue; }
        if (all_cells) c[u] = i;
        else if (i < n_uniform) c[u] = min((uint32_t)(u_cell[i] * (float)n_total_cells), n_total_cells - 1u);
        else {
            const uint32_t j = i - n_uniform;
            c[u] = n_occ > n_uniform ? occupied_cells[min((uint32_t)(u_pick[j] * (float)n_occ), n_occ - 1u)] : occupied_cells[j];
        }
        const uint32_t cz = c[u] % res.z, cy = (c[u] / res.z) % res.y, cx = c[u] / (res.z * res.y);
        xu[u][0] = ((float)cx + jitter[3ull * i]) / (float)res.x;
        xu[u][1] = ((float)cy + jitter[3ull * i + 1]) / (float)res.y;
        xu[u][2] = ((float)cz + jitter[3ull * i + 2]) / (float)res.z;
        const float a = xu[u][0] - 0.5f, b = xu[u][1] - 0.5f, d = xu[u][2] - 0.5f;
        if (!sphere || sqrtf(a * a + b * b + d * d) < 0.5f) { keep |= 1u << u; ++n_keep; }
    }
    uint32_t o = i0;
    if (sphere) {  // exclusive prefix of n_keep over the workgroup (in slot order) + the workgroup's place in the output
        uint32_t v = n_keep;
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) {
            const uint32_t t = __shfl_up(v, k, 64);
            if (lane >= k) v += t;
        }
        if (lane == 63) wave_tot[w] = v;
        __syncthreads();
        uint32_t before = 0, all = 0;
#pragma unroll
        for (int k = 0; k < EW_BLOCK / 64; ++k) {
            if (k < w) before += wave_tot[k];
            all += wave_tot[k];
        }
        if (threadIdx.x == 0) block_base = all ? (uint32_t)atomicAdd(n_cells, (int32_t)all) : 0u;
        __syncthreads();
        o = block_base + before + v - n_keep;
    }
#pragma unroll
    for (int u = 0; u < OCC_SPT; ++u) {
        if (!(keep & (1u << u))) continue;
        const uint32_t q = sphere ? o++ : i0 + u;
        cells[q] = c[u];
        x_unit[3ull * q] = xu[u][0];
        x_unit[3ull * q + 1] = xu[u][1];
        x_unit[3ull * q + 2] = xu[u][2];
    }
}

// occ = trunc_exp(logit + bias) * step ;  occs_new[c] = max(occs_old[c] * decay, occ)
__global__ void __launch_bounds__(EW_BLOCK)
k_occ_ema(const __half *__restrict__ mlp_out, uint32_t stride, float bias, float step, float decay,
          const uint32_t *__restrict__ cells, const float *__restrict__ occs_old, float *__restrict__ occs_new,
          uint32_t capacity, const int32_t *__restrict__ n_cells)
{
    const uint32_t i = blockIdx.x * EW_BLOCK + threadIdx.x;
    if (i >= live_count(capacity, n_cells)) return;
    const float occ = expf(__half2float(mlp_out[(uint64_t)i * stride]) + bias) * step;
    const uint32_t c = cells[i];
    occs_new[c] = fmaxf(occs_old[c] * decay, occ);  // duplicates: one of them wins, all computed from the OLD value
}

// ... with the per-cell occupancy statistic already evaluated by the caller (NeuS: the closed-form alpha of one step)
__global__ void __launch_bounds__(EW_BLOCK)
k_occ_ema_values(const float *__restrict__ occ, float decay, const uint32_t *__restrict__ cells,
                 const float *__restrict__ occs_old, float *__restrict__ occs_new, uint32_t capacity,
                 const int32_t *__restrict__ n_cells)
{
    const uint32_t i = blockIdx.x * EW_BLOCK + threadIdx.x;
    if (i >= live_count(capacity, n_cells)) return;
    const float v = occ[i];
    if (v < 0.f) return;  // the caller's "do not touch this cell" (a sample outside the unit sphere of a contracted grid)
    const uint32_t c = cells[i];
    occs_new[c] = fmaxf(occs_old[c] * decay, v);
}

// NeRF++ background grid (models/neus.py:103-106, nerfacc 0.3.3 _update with UN_BOUNDED_SPHERE): occ = exp(logit + bias) *
// step for samples inside the unit sphere of the grid's contracted space, -1 ("leave the cell alone": nerfacc drops such
// samples before it evaluates them) outside
__global__ void __launch_bounds__(EW_BLOCK)
k_occ_density_values_sphere(const float *__restrict__ logit, const float *__restrict__ x_unit, float bias, float step,
                            float *__restrict__ occ, uint32_t capacity, const int32_t *__restrict__ n_cells)
{
    const uint32_t i = blockIdx.x * EW_BLOCK + threadIdx.x;
    if (i >= live_count(capacity, n_cells)) return;
    const float a = x_unit[3ull * i] - 0.5f, b = x_unit[3ull * i + 1] - 0.5f, c = x_unit[3ull * i + 2] - 0.5f;
    const bool inside = sqrtf(a * a + b * b + c * c) < 0.5f;
    occ[i] = inside ? expf(logit[i] + bias) * step : -1.f;
}

constexpr int OCC_PARTS = 256;

// mean(occs) in two steps: OCC_PARTS partial sums (a single workgroup walking 8 MB serially took ~1 ms) ...
__global__ void __launch_bounds__(EW_BLOCK)
k_occ_partial_sums(const float *__restrict__ occs, uint32_t n, double *__restrict__ partial)
{
    __shared__ double part[EW_BLOCK / 64];
    double s = 0.0;
    for (uint32_t k = blockIdx.x * EW_BLOCK + threadIdx.x; k < n; k += OCC_PARTS * EW_BLOCK) s += (double)occs[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < EW_BLOCK / 64; ++w) t += part[w];
        partial[blockIdx.x] = t;
    }
}

// ... summed again by every workgroup of the binarisation: threshold = min(mean, occ_thre) = torch.clamp(occs.mean(), max=)
__global__ void __launch_bounds__(EW_BLOCK)
k_occ_binarize(const float *__restrict__ occs, const double *__restrict__ partial, float occ_thre,
               float *__restrict__ thr_out, uint8_t *__restrict__ binary, uint32_t n)
{
    __shared__ float thr_s;
    if (threadIdx.x < 64) {
        double s = 0.0;
        for (int k = threadIdx.x; k < OCC_PARTS; k += 64) s += partial[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (threadIdx.x == 0) {
            thr_s = fminf((float)(s / (double)n), occ_thre);
            if (blockIdx.x == 0) thr_out[0] = thr_s;
        }
    }
    __syncthreads();
    const uint32_t i = blockIdx.x * EW_BLOCK + threadIdx.x;
    if (i < n) binary[i] = occs[i] > thr_s ? 1 : 0;
}

}  // namespace

extern "C" int nsr_occupancy_select_cells(const uint64_t *bricks, int res_x, int res_y, int res_z, const float *u_cell,
                                          const float *u_pick, const float *jitter, uint32_t n_uniform, int all_cells,
                                          uint32_t capacity, uint32_t *brick_offset, uint32_t *occupied_cells,
                                          int32_t *n_occupied, uint32_t *cells, float *x_unit, int32_t *n_cells,
                                          void *stream)
{
    NSR_REQUIRE(res_x > 0 && res_y > 0 && res_z > 0 && (res_x & 3) == 0 && (res_y & 3) == 0 && (res_z & 3) == 0,
                "nsr_occupancy_select_cells: resolution must be a multiple of 4");
    const int sphere = (all_cells >> 1) & 1;  // bit 1 of the flag word: the grid lives on a sphere-contracted space
    all_cells &= 1;
    const uint32_t n_bricks = (uint32_t)((res_x >> 2) * (res_y >> 2) * (res_z >> 2));
    const uint64_t n_total = (uint64_t)res_x * res_y * res_z;
    NSR_REQUIRE(n_total < (1ull << 31), "nsr_occupancy_select_cells: grid too large");
    NSR_REQUIRE(jitter && cells && x_unit && n_cells, "nsr_occupancy_select_cells: NULL pointer");
    NSR_REQUIRE(all_cells ? capacity >= n_total : (bricks && u_cell && u_pick && brick_offset && occupied_cells &&
                                                   n_occupied && capacity >= 2ull * n_uniform),
                "nsr_occupancy_select_cells: missing buffer or capacity too small");
    hipStream_t st = (hipStream_t)stream;
    const int3 res = make_int3(res_x, res_y, res_z);
    if (!all_cells) {
        hipLaunchKernelGGL(k_occ_brick_scan, dim3(1), dim3(1024), 0, st, (const unsigned long long *)bricks, n_bricks,
                           brick_offset, n_occupied);
        hipLaunchKernelGGL(k_occ_brick_expand, dim3(nsr_div_up(n_bricks, EW_BLOCK)), dim3(EW_BLOCK), 0, st,
                           (const unsigned long long *)bricks, brick_offset, res, n_bricks, occupied_cells);
    }
    const uint32_t launch = all_cells ? (uint32_t)n_total : 2u * n_uniform;
    if (sphere)
        NSR_REQUIRE(hipMemsetAsync(n_cells, 0, sizeof(int32_t), st) == hipSuccess, "nsr_occupancy_select_cells: memset failed");
    hipLaunchKernelGGL(k_occ_make_samples, dim3(nsr_div_up(launch, EW_BLOCK * OCC_SPT)), dim3(EW_BLOCK), 0, st, occupied_cells,
                       n_occupied, u_cell, u_pick, jitter, res, n_uniform, all_cells, sphere, capacity, cells, x_unit, n_cells);
    NSR_CHECK_LAUNCH("nsr_occupancy_select_cells");
    return NSR_OK;
}

extern "C" int nsr_occupancy_update(const nsr_half *mlp_out, uint32_t stride, float density_bias, float step_size,
                                    float ema_decay, float occ_thre, const uint32_t *cells, const float *occs_old,
                                    float *occs_new, uint8_t *binary, float *threshold, uint32_t n_total_cells,
                                    uint32_t capacity, const int32_t *n_cells, void *stream)
{
    NSR_REQUIRE(mlp_out && cells && occs_old && occs_new && binary && threshold && n_cells && occs_old != occs_new,
                "nsr_occupancy_update: NULL pointer (occs_old and occs_new must be distinct buffers)");
    NSR_REQUIRE(((uintptr_t)threshold & 7u) == 0, "nsr_occupancy_update: threshold workspace must be 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    NSR_REQUIRE(hipMemcpyAsync(occs_new, occs_old, (size_t)n_total_cells * sizeof(float), hipMemcpyDeviceToDevice, st) ==
                    hipSuccess, "nsr_occupancy_update: hipMemcpyAsync failed");
    hipLaunchKernelGGL(k_occ_ema, dim3(nsr_div_up(capacity, EW_BLOCK)), dim3(EW_BLOCK), 0, st, (const __half *)mlp_out,
                       stride, density_bias, step_size, ema_decay, cells, occs_old, occs_new, capacity, n_cells);
    double *partial = reinterpret_cast<double *>(threshold + 2);  // threshold: [0] the value, [2 ..] the partial sums
    hipLaunchKernelGGL(k_occ_partial_sums, dim3(OCC_PARTS), dim3(EW_BLOCK), 0, st, occs_new, n_total_cells, partial);
    hipLaunchKernelGGL(k_occ_binarize, dim3(nsr_div_up(n_total_cells, EW_BLOCK)), dim3(EW_BLOCK), 0, st, occs_new,
                       partial, occ_thre, threshold, binary, n_total_cells);
    NSR_CHECK_LAUNCH("nsr_occupancy_update");
    return NSR_OK;
}

extern "C" int nsr_occupancy_density_values_sphere(const float *logit, const float *x_unit, float density_bias,
                                                   float step_size, float *occ, uint32_t capacity, const int32_t *n_cells,
                                                   void *stream)
{
    NSR_REQUIRE(logit && x_unit && occ && n_cells, "nsr_occupancy_density_values_sphere: NULL pointer");
    if (capacity == 0) return NSR_OK;
    hipLaunchKernelGGL(k_occ_density_values_sphere, dim3(nsr_div_up(capacity, EW_BLOCK)), dim3(EW_BLOCK), 0,
                       (hipStream_t)stream, logit, x_unit, density_bias, step_size, occ, capacity, n_cells);
    NSR_CHECK_LAUNCH("nsr_occupancy_density_values_sphere");
    return NSR_OK;
}

extern "C" int nsr_occupancy_update_values(const float *occ_values, float ema_decay, float occ_thre, const uint32_t *cells,
                                           const float *occs_old, float *occs_new, uint8_t *binary, float *threshold,
                                           uint32_t n_total_cells, uint32_t capacity, const int32_t *n_cells, void *stream)
{
    NSR_REQUIRE(occ_values && cells && occs_old && occs_new && binary && threshold && n_cells,
                "nsr_occupancy_update_values: NULL pointer");
    NSR_REQUIRE(((uintptr_t)threshold & 7u) == 0, "nsr_occupancy_update_values: threshold must be 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    NSR_REQUIRE(hipMemcpyAsync(occs_new, occs_old, (size_t)n_total_cells * sizeof(float), hipMemcpyDeviceToDevice, st) ==
                    hipSuccess, "nsr_occupancy_update_values: copy failed");
    if (capacity > 0)
        hipLaunchKernelGGL(k_occ_ema_values, dim3(nsr_div_up(capacity, EW_BLOCK)), dim3(EW_BLOCK), 0, st, occ_values,
                           ema_decay, cells, occs_old, occs_new, capacity, n_cells);
    double *partial = reinterpret_cast<double *>(threshold + 2);
    hipLaunchKernelGGL(k_occ_partial_sums, dim3(OCC_PARTS), dim3(EW_BLOCK), 0, st, occs_new, n_total_cells, partial);
    hipLaunchKernelGGL(k_occ_binarize, dim3(nsr_div_up(n_total_cells, EW_BLOCK)), dim3(EW_BLOCK), 0, st, occs_new,
                       partial, occ_thre, threshold, binary, n_total_cells);
    NSR_CHECK_LAUNCH("nsr_occupancy_update_values");
    return NSR_OK;
}
