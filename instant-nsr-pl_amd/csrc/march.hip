// Occupancy-grid ray marching for gfx950 (replaces nerfacc 0.3.3 ray_aabb_intersect / ray_marching /
// contraction / grid query; reference call sites models/nerf.py:83, models/neus.py:153,159,210).
//
// Parity contract: ray_indices / packed_info / sample counts are BIT-EXACT against the oracle
// (oracle/csrc/nerfacc_ref.c) and t_starts / t_ends are bit-exact fp32: every arithmetic op below is a
// separately rounded IEEE op (this translation unit is built with -ffp-contract=off; hipcc's default
// fp32 division and sqrt are correctly rounded), except the sample position  p = fmaf(t, d, o).
//
// MI355X notes: one lane = one ray (the step recurrence is sequential in fp32 by definition); rays of a
// training batch are drawn i.i.d. so trip counts diverge inside a wavefront -- the kernel is latency /
// divergence bound, not bandwidth bound (a 128^3 bool grid is 2 MiB and stays in the XCD's L2).  The
// two passes (count, write) share one code path through a template flag so both see identical floats.
#include <math.h>

#include "nsr_common.h"

#pragma clang fp contract(off)

namespace {

constexpr int MARCH_BLOCK = 128;  // 2 waves/block: more blocks in flight for the divergent loop
constexpr int EW_BLOCK = 256;

struct Roi { float lo[3], hi[3]; };

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
__device__ __forceinline__ float signf(float v) { return (float)((v > 0.f) - (v < 0.f)); }

__device__ __forceinline__ void aabb_one(const float *o, const float *d, const float *aabb, float &near, float &far)
{
    float tmin = (aabb[0] - o[0]) / d[0], tmax = (aabb[3] - o[0]) / d[0];
    if (tmin > tmax) { float s = tmin; tmin = tmax; tmax = s; }
    float tymin = (aabb[1] - o[1]) / d[1], tymax = (aabb[4] - o[1]) / d[1];
    if (tymin > tymax) { float s = tymin; tymin = tymax; tymax = s; }
    if (tmin > tymax || tymin > tmax) { near = 1e10f; far = 1e10f; return; }
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = (aabb[2] - o[2]) / d[2], tzmax = (aabb[5] - o[2]) / d[2];
    if (tzmin > tzmax) { float s = tzmin; tzmin = tzmax; tzmax = s; }
    if (tmin > tzmax || tzmin > tmax) { near = 1e10f; far = 1e10f; return; }
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    near = tmin > 0.f ? tmin : 0.f;
    far = tmax;
}

__global__ void __launch_bounds__(EW_BLOCK)
k_ray_aabb(const float *__restrict__ rays_o, const float *__restrict__ rays_d, const float *__restrict__ aabb,
           float *__restrict__ t_min, float *__restrict__ t_max, uint32_t n)
{
    const uint32_t i = blockIdx.x * EW_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float o[3] = {rays_o[3ull * i], rays_o[3ull * i + 1], rays_o[3ull * i + 2]};
    const float d[3] = {rays_d[3ull * i], rays_d[3ull * i + 1], rays_d[3ull * i + 2]};
    const float bb[6] = {aabb[0], aabb[1], aabb[2], aabb[3], aabb[4], aabb[5]};
    float a, b;
    aabb_one(o, d, bb, a, b);
    t_min[i] = a;
    t_max[i] = b;
}

__device__ __forceinline__ void roi_to_unit(const float *p, const Roi &r, float *u)
{
#pragma unroll
    for (int k = 0; k < 3; ++k) u[k] = (p[k] - r.lo[k]) / (r.hi[k] - r.lo[k]);
}

__device__ __forceinline__ void apply_contraction(const float *p, const Roi &r, int type, float *u)
{
    roi_to_unit(p, r, u);
    if (type == NSR_CONTRACT_UN_BOUNDED_SPHERE) {
        float v[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] = u[k] * 2.f - 1.f;
        const float n = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        if (n > 1.f) {
            const float s = 2.f - 1.f / n;
#pragma unroll
            for (int k = 0; k < 3; ++k) v[k] = s * (v[k] / n);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) u[k] = v[k] * 0.25f + 0.5f;
    } else if (type == NSR_CONTRACT_UN_BOUNDED_TANH) {
#pragma unroll
        for (int k = 0; k < 3; ++k) u[k] = tanhf((u[k] - 0.5f) * 1.0986122886681098f) * 0.5f + 0.5f;
    }
}

__device__ __forceinline__ void apply_contraction_inv(const float *u, const Roi &r, int type, float *p)
{
    float w[3] = {u[0], u[1], u[2]};
    if (type == NSR_CONTRACT_UN_BOUNDED_SPHERE) {
        float v[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] = (u[k] - 0.5f) * 4.f;
        const float n = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        if (n > 1.f) {
            const float s = 1.f / (2.f - n);
#pragma unroll
            for (int k = 0; k < 3; ++k) v[k] = (v[k] / n) * s;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) w[k] = v[k] * 0.5f + 0.5f;
    } else if (type == NSR_CONTRACT_UN_BOUNDED_TANH) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float t = clampf((u[k] - 0.5f) * 2.f, -1.f + 1e-6f, 1.f - 1e-6f);
            w[k] = atanhf(t) / 1.0986122886681098f + 0.5f;
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) p[k] = w[k] * (r.hi[k] - r.lo[k]) + r.lo[k];
}

__device__ __forceinline__ int grid_idx_at(const float *u, const int3 res)
{
    int ix = (int)(u[0] * (float)res.x), iy = (int)(u[1] * (float)res.y), iz = (int)(u[2] * (float)res.z);
    ix = min(max(ix, 0), res.x - 1);
    iy = min(max(iy, 0), res.y - 1);
    iz = min(max(iz, 0), res.z - 1);
    return ix * res.y * res.z + iy * res.z + iz;
}

__device__ __forceinline__ bool outside_roi(const float *p, const Roi &r)
{
    return p[0] < r.lo[0] || p[0] > r.hi[0] || p[1] < r.lo[1] || p[1] > r.hi[1] || p[2] < r.lo[2] || p[2] > r.hi[2];
}

__device__ __forceinline__ bool grid_occupied_at(const float *p, const Roi &r, int type, const int3 res,
                                                 const uint8_t *__restrict__ grid)
{
    if (type == NSR_CONTRACT_AABB && outside_roi(p, r)) return false;
    float u[3];
    apply_contraction(p, r, type, u);
    return grid[grid_idx_at(u, res)] != 0;
}

__device__ __forceinline__ float calc_dt(float t, float cone_angle, float dt_min, float dt_max)
{
    return clampf(t * cone_angle, dt_min, dt_max);
}

__device__ __forceinline__ float distance_to_next_voxel(const float *p, const float *d, const float *inv_d,
                                                        const Roi &r, const int3 res)
{
    float u[3];
    roi_to_unit(p, r, u);
    const float rr[3] = {(float)res.x, (float)res.y, (float)res.z};
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float x = u[k] * rr[k];
        const float tk = ((floorf(x + 0.5f + 0.5f * signf(d[k])) - x) * inv_d[k]) / rr[k] * (r.hi[k] - r.lo[k]);
        t = (k == 0) ? tk : fminf(t, tk);
    }
    return fmaxf(t, 0.f);
}

template <bool WRITE>
__global__ void __launch_bounds__(MARCH_BLOCK)
k_ray_march(const float *__restrict__ rays_o, const float *__restrict__ rays_d, const float *__restrict__ t_min,
            const float *__restrict__ t_max, const float *__restrict__ roi, const uint8_t *__restrict__ grid,
            int3 res, int type, float step, float cone_angle, const int32_t *__restrict__ packed_info,
            int32_t *__restrict__ num_steps, int64_t *__restrict__ ray_indices, float *__restrict__ t_starts,
            float *__restrict__ t_ends, uint32_t n_rays)
{
    const uint32_t i = blockIdx.x * MARCH_BLOCK + threadIdx.x;
    if (i >= n_rays) return;
    Roi r;
#pragma unroll
    for (int k = 0; k < 3; ++k) { r.lo[k] = roi[k]; r.hi[k] = roi[3 + k]; }
    const float o[3] = {rays_o[3ull * i], rays_o[3ull * i + 1], rays_o[3ull * i + 2]};
    const float d[3] = {rays_d[3ull * i], rays_d[3ull * i + 1], rays_d[3ull * i + 2]};
    const float inv_d[3] = {1.f / d[0], 1.f / d[1], 1.f / d[2]};
    const float near = t_min[i], far = t_max[i];
    const float dt_min = step, dt_max = 1e10f;
    int64_t base = 0;
    if (WRITE) base = packed_info[2ull * i];

    int j = 0;
    float t0 = near;
    float dt = calc_dt(t0, cone_angle, dt_min, dt_max);
    float t1 = t0 + dt;
    float t_mid = (t0 + t1) * 0.5f;
    while (t_mid < far) {
        const float p[3] = {__builtin_fmaf(t_mid, d[0], o[0]), __builtin_fmaf(t_mid, d[1], o[1]),
                            __builtin_fmaf(t_mid, d[2], o[2])};
        if (grid_occupied_at(p, r, type, res, grid)) {
            if (WRITE) {
                t_starts[base + j] = t0;
                t_ends[base + j] = t1;
                ray_indices[base + j] = (int64_t)i;
            }
            ++j;
            t0 = t1;
            t1 = t0 + calc_dt(t0, cone_angle, dt_min, dt_max);
            t_mid = (t0 + t1) * 0.5f;
        } else if (type == NSR_CONTRACT_AABB) {
            const float t_target = t_mid + distance_to_next_voxel(p, d, inv_d, r, res);
            do { t_mid += dt_min; } while (t_mid < t_target);
            dt = calc_dt(t_mid, cone_angle, dt_min, dt_max);
            t0 = t_mid - dt * 0.5f;
            t1 = t_mid + dt * 0.5f;
        } else {
            t0 = t1;
            t1 = t0 + calc_dt(t0, cone_angle, dt_min, dt_max);
            t_mid = (t0 + t1) * 0.5f;
        }
    }
    if (!WRITE) num_steps[i] = j;
}

// ------------------------------------------------------------------------------------------------
// Brick-packed occupancy grid + single-pass marching (the default path).
//
// Profile of the byte-grid two-pass marcher at 8192 rays: 2 x 445 us per step, 30 % of the whole training step.
// It is a pure dependent-load chain: every voxel visit waits ~700 clk on an L2 byte load and only 128 wavefronts
// exist.  Same booleans, same fp32 recurrence, far fewer and cheaper loads:
//   * the bool grid is re-packed into 4x4x4 BRICKS of 64 bits (one 8-B load serves ~5 voxel visits / ~20 steps)
//     plus one "any voxel set" bit per brick; the any-bits (4 KiB for 128^3, 32 KiB for 256^3) are staged in LDS,
//     so empty space is skipped without touching global memory at all;
//   * ONE marching pass writes (t0,t1) into a per-ray scratch row of the provable capacity, then a wave-per-ray
//     copy packs them: the loop runs once instead of twice.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_pack_bricks(const uint8_t *__restrict__ grid, int3 res, unsigned long long *__restrict__ bricks,
              uint32_t *__restrict__ any_bits, uint32_t n_bricks)
{
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    unsigned long long bits = 0ull;
    if (b < n_bricks) {
        const int nby = res.y >> 2, nbz = res.z >> 2;
        const int bz = b % nbz, by = (b / nbz) % nby, bx = b / (nbz * nby);
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                const uint32_t v = *reinterpret_cast<const uint32_t *>(
                    grid + ((size_t)(bx * 4 + i) * res.y + (by * 4 + j)) * res.z + bz * 4);
                for (int k = 0; k < 4; ++k)
                    if ((v >> (8 * k)) & 0xffu) bits |= 1ull << ((i * 4 + j) * 4 + k);
            }
        bricks[b] = bits;
    }
    // one any-bit per brick, 32 bricks per word (wave ballot: lanes 0-31 -> word 2w, lanes 32-63 -> word 2w+1)
    const unsigned long long m = __ballot(bits != 0ull);
    const uint32_t lane = threadIdx.x & 63;
    if (lane == 0 && b < n_bricks) any_bits[b >> 5] = (uint32_t)m;
    if (lane == 32 && b < n_bricks) any_bits[b >> 5] = (uint32_t)(m >> 32);
}

// contraction applied to already-computed unit coordinates u = roi_to_unit(p) (same float ops as apply_contraction)
__device__ __forceinline__ void contract_unit(float *u, int type)
{
    if (type == NSR_CONTRACT_UN_BOUNDED_SPHERE) {
        float v[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) v[k] = u[k] * 2.f - 1.f;
        const float n = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        if (n > 1.f) {
            const float s = 2.f - 1.f / n;
#pragma unroll
            for (int k = 0; k < 3; ++k) v[k] = s * (v[k] / n);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) u[k] = v[k] * 0.25f + 0.5f;
    } else if (type == NSR_CONTRACT_UN_BOUNDED_TANH) {
#pragma unroll
        for (int k = 0; k < 3; ++k) u[k] = tanhf((u[k] - 0.5f) * 1.0986122886681098f) * 0.5f + 0.5f;
    }
}

// distance_to_next_voxel on precomputed unit coordinates.  POW2: the grid resolution is a power of two, so the
// division by it is exactly a multiplication by its reciprocal (identical bits, one instruction instead of ~10).
template <bool POW2>
__device__ __forceinline__ float distance_to_next_voxel_u(const float *u, const float *d, const float *inv_d,
                                                          const Roi &r, const float *rr, const float *inv_rr)
{
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float x = u[k] * rr[k];
        const float a = (floorf(x + 0.5f + 0.5f * signf(d[k])) - x) * inv_d[k];
        const float tk = (POW2 ? a * inv_rr[k] : a / rr[k]) * (r.hi[k] - r.lo[k]);
        t = (k == 0) ? tk : fminf(t, tk);
    }
    return fmaxf(t, 0.f);
}

struct BrickGrid {
    const unsigned long long *bricks;
    const uint32_t *any_lds;  // LDS copy of the any-bits
    int3 res;
    int nby, nbz;
    uint32_t cur_id;
    unsigned long long cur_bits;
};

// occupancy test on precomputed unit coordinates `u` (already contracted)
__device__ __forceinline__ bool brick_occupied_u(const float *u, BrickGrid &bg)
{
    int ix = (int)(u[0] * (float)bg.res.x), iy = (int)(u[1] * (float)bg.res.y), iz = (int)(u[2] * (float)bg.res.z);
    ix = min(max(ix, 0), bg.res.x - 1);
    iy = min(max(iy, 0), bg.res.y - 1);
    iz = min(max(iz, 0), bg.res.z - 1);
    const uint32_t id = (uint32_t)(((ix >> 2) * bg.nby + (iy >> 2)) * bg.nbz + (iz >> 2));
    if (id != bg.cur_id) {
        bg.cur_id = id;
        bg.cur_bits = ((bg.any_lds[id >> 5] >> (id & 31u)) & 1u) ? bg.bricks[id] : 0ull;
    }
    return (bg.cur_bits >> (((ix & 3) * 4 + (iy & 3)) * 4 + (iz & 3))) & 1ull;
}

// MODE 0: count only (num_steps) ; MODE 1: write at packed_info ; MODE 2: single pass into scratch rows of `cap`
template <int MODE, bool POW2>
__global__ void __launch_bounds__(MARCH_BLOCK)
k_ray_march_bricks(const float *__restrict__ rays_o, const float *__restrict__ rays_d, const float *__restrict__ t_min,
                   const float *__restrict__ t_max, const float *__restrict__ roi,
                   const unsigned long long *__restrict__ bricks, const uint32_t *__restrict__ any_bits,
                   uint32_t n_any_words, int3 res, int type, float step, float cone_angle,
                   const int32_t *__restrict__ packed_info, int32_t *__restrict__ num_steps,
                   int64_t *__restrict__ ray_indices, float *__restrict__ t_starts, float *__restrict__ t_ends,
                   float2 *__restrict__ scratch, uint32_t cap, uint32_t n_rays, uint32_t rays_per_wave)
{
    extern __shared__ uint32_t any_lds[];
    for (uint32_t k = threadIdx.x; k < n_any_words; k += MARCH_BLOCK) any_lds[k] = any_bits[k];
    __syncthreads();
    // A wave takes as long as its slowest ray (up to ~1,000 dependent visits) and pays for both sides of every branch its
    // lanes disagree on; 8,192 rays are 128 full waves on a chip with 1,024 SIMDs.  So a launch of few rays spreads them:
    // only the first `rays_per_wave` lanes of a wave carry a ray (launch_bricks picks 64 / 16 / 8 by the ray count).
    const uint32_t lane = threadIdx.x & 63u;
    if (lane >= rays_per_wave) return;
    const uint32_t i = ((blockIdx.x * MARCH_BLOCK + threadIdx.x) >> 6) * rays_per_wave + lane;
    if (i >= n_rays) return;
    Roi r;
#pragma unroll
    for (int k = 0; k < 3; ++k) { r.lo[k] = roi[k]; r.hi[k] = roi[3 + k]; }
    const float o[3] = {rays_o[3ull * i], rays_o[3ull * i + 1], rays_o[3ull * i + 2]};
    const float d[3] = {rays_d[3ull * i], rays_d[3ull * i + 1], rays_d[3ull * i + 2]};
    const float inv_d[3] = {1.f / d[0], 1.f / d[1], 1.f / d[2]};
    const float near = t_min[i], far = t_max[i];
    const float dt_min = step, dt_max = 1e10f;
    BrickGrid bg{bricks, any_lds, res, res.y >> 2, res.z >> 2, 0xffffffffu, 0ull};
    int64_t base = 0;
    if (MODE == 1) base = packed_info[2ull * i];
    float2 *row = (MODE == 2) ? scratch + (uint64_t)i * cap : nullptr;

    uint32_t j = 0;
    float t0 = near;
    float dt = calc_dt(t0, cone_angle, dt_min, dt_max);
    float t1 = t0 + dt;
    float t_mid = (t0 + t1) * 0.5f;
    const float rr[3] = {(float)res.x, (float)res.y, (float)res.z};
    const float inv_rr[3] = {1.f / rr[0], 1.f / rr[1], 1.f / rr[2]};
    while (t_mid < far) {
        const float p[3] = {__builtin_fmaf(t_mid, d[0], o[0]), __builtin_fmaf(t_mid, d[1], o[1]),
                            __builtin_fmaf(t_mid, d[2], o[2])};
        // unit coordinates once per visit: shared by the occupancy test and the voxel-exit distance
        float u[3], uc[3];
        roi_to_unit(p, r, u);
        bool occ = false;
        if (!(type == NSR_CONTRACT_AABB && outside_roi(p, r))) {
            uc[0] = u[0]; uc[1] = u[1]; uc[2] = u[2];
            contract_unit(uc, type);
            occ = brick_occupied_u(uc, bg);
        }
        if (occ) {
            if (MODE == 1) {
                t_starts[base + j] = t0;
                t_ends[base + j] = t1;
                ray_indices[base + j] = (int64_t)i;
            } else if (MODE == 2) {
                if (j < cap) row[j] = make_float2(t0, t1);
            }
            ++j;
            t0 = t1;
            t1 = t0 + calc_dt(t0, cone_angle, dt_min, dt_max);
            t_mid = (t0 + t1) * 0.5f;
        } else if (type == NSR_CONTRACT_AABB) {
            const float t_target = t_mid + distance_to_next_voxel_u<POW2>(u, d, inv_d, r, rr, inv_rr);
            do { t_mid += dt_min; } while (t_mid < t_target);
            dt = calc_dt(t_mid, cone_angle, dt_min, dt_max);
            t0 = t_mid - dt * 0.5f;
            t1 = t_mid + dt * 0.5f;
        } else {
            t0 = t1;
            t1 = t0 + calc_dt(t0, cone_angle, dt_min, dt_max);
            t_mid = (t0 + t1) * 0.5f;
        }
    }
    if (MODE != 1) num_steps[i] = (int32_t)j;
}

// ------------------------------------------------------------------------------------------------
// WAVE-PER-RAY marcher (AABB contraction, cone_angle == 0: the bounded scenes of every reference config).
//
// The lane-per-ray kernel above is one dependent chain per ray -- position, three divisions, voxel index, brick word, bit
// test, next t: ~700 clk per visit, up to ~1,000 visits -- and a launch lasts as long as its longest ray: 360 us for 8,192
// rays whether they sit 64 or 4 to a wave (tools/march_bench.py), with 7/8 of the chip idle.  What is sequential in
// nerfacc's loop is only the fp32 recurrence of t; WHICH t values get visited depends on occupancy, but the candidates do
// not:
//   * inside an occupied stretch the samples are t0' = t1, t1' = t0' + dt: lane j takes the j-th of the next 64 (every lane
//     runs the 64 adds, they are cheap), all 64 occupancy tests -- the expensive part -- run at once, a ballot finds the
//     first sample that is empty or beyond t_max, the lanes in front of it store their samples;
//   * in empty space the loop visits a SUBSEQUENCE of b' = b + dt: lane n takes b_n, tests it and computes the voxel-exit
//     distance; "add dt until t >= t_mid + distance" is a search for the first b_m >= target among the 64 candidates (six
//     rounds of lane shuffles for all lanes at once), and the loop's walk through the window is a chase over those
//     indices with wave-uniform lane reads.
// Every float is produced by the same operations in the same order as in the serial loop: the outputs are bit-identical
// (tests/test_gpu_march.py compares both kernels with the C oracle).
// ------------------------------------------------------------------------------------------------
constexpr int MARCH_WAVE_BLOCK = 256;  // 4 rays per workgroup (they share the LDS copy of the any-bits)

template <int MODE, bool POW2>
__global__ void __launch_bounds__(MARCH_WAVE_BLOCK)
k_ray_march_wave(const float *__restrict__ rays_o, const float *__restrict__ rays_d, const float *__restrict__ t_min,
                 const float *__restrict__ t_max, const float *__restrict__ roi,
                 const unsigned long long *__restrict__ bricks, const uint32_t *__restrict__ any_bits,
                 uint32_t n_any_words, int3 res, float step, const int32_t *__restrict__ packed_info,
                 int32_t *__restrict__ num_steps, int64_t *__restrict__ ray_indices, float *__restrict__ t_starts,
                 float *__restrict__ t_ends, float2 *__restrict__ scratch, uint32_t cap, uint32_t n_rays)
{
    extern __shared__ uint32_t any_lds[];
    for (uint32_t k = threadIdx.x; k < n_any_words; k += MARCH_WAVE_BLOCK) any_lds[k] = any_bits[k];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t i = (blockIdx.x * MARCH_WAVE_BLOCK + threadIdx.x) >> 6;
    if (i >= n_rays) return;  // (wave-uniform)
    Roi r;
#pragma unroll
    for (int k = 0; k < 3; ++k) { r.lo[k] = roi[k]; r.hi[k] = roi[3 + k]; }
    const float o[3] = {rays_o[3ull * i], rays_o[3ull * i + 1], rays_o[3ull * i + 2]};
    const float d[3] = {rays_d[3ull * i], rays_d[3ull * i + 1], rays_d[3ull * i + 2]};
    const float inv_d[3] = {1.f / d[0], 1.f / d[1], 1.f / d[2]};
    const float near = t_min[i], far = t_max[i];
    const float dt = calc_dt(near, 0.f, step, 1e10f);  // cone_angle 0: the step size, always
    const float half = dt * 0.5f;
    const float rr[3] = {(float)res.x, (float)res.y, (float)res.z};
    const float inv_rr[3] = {1.f / rr[0], 1.f / rr[1], 1.f / rr[2]};
    const int nby = res.y >> 2, nbz = res.z >> 2;
    int64_t base = 0;
    if (MODE == 1) base = packed_info[2ull * i];
    float2 *row = (MODE == 2) ? scratch + (uint64_t)i * cap : nullptr;

    // occupancy of the sample at t (+ the distance to the voxel's exit): the visit of the serial loop, per lane
    auto visit = [&](float t, float &dist) -> bool {
        const float p[3] = {__builtin_fmaf(t, d[0], o[0]), __builtin_fmaf(t, d[1], o[1]), __builtin_fmaf(t, d[2], o[2])};
        float u[3];
        roi_to_unit(p, r, u);
        bool occ = false;
        if (!outside_roi(p, r)) {
            int ix = (int)(u[0] * rr[0]), iy = (int)(u[1] * rr[1]), iz = (int)(u[2] * rr[2]);
            ix = min(max(ix, 0), res.x - 1);
            iy = min(max(iy, 0), res.y - 1);
            iz = min(max(iz, 0), res.z - 1);
            const uint32_t id = (uint32_t)(((ix >> 2) * nby + (iy >> 2)) * nbz + (iz >> 2));
            if ((any_lds[id >> 5] >> (id & 31u)) & 1u)
                occ = (bricks[id] >> (((ix & 3) * 4 + (iy & 3)) * 4 + (iz & 3))) & 1ull;
        }
        dist = distance_to_next_voxel_u<POW2>(u, d, inv_d, r, rr, inv_rr);
        return occ;
    };

    // n sequential fp32 additions of dt starting from a (a > 0) stay on an arithmetic progression of BIT PATTERNS while the
    // sums stay inside a's binade: a = m u (u the binade's ulp, m an integer), the exact sum m u + dt rounds to (m + k) u with
    // k = round(dt / u) whenever dt / u is not an exact tie -- the same k for every m.  -> true and k_ulp when a + n adds can
    // be taken that way (wave-uniform: a is); otherwise the caller runs the additions.
    auto progression = [&](float a, int n, int &k_ulp) -> bool {
        const int bits = __float_as_int(a);
        if (!(a > 0.f) || (bits >> 23) == 0 || (bits >> 23) >= 254) return false;
        const float ulp = __int_as_float(((bits >> 23) - 23) << 23);  // 2^(e - 23)  (e - 23 > 0 for every t a scene produces)
        if (((bits >> 23) - 23) <= 0) return false;
        const float x = dt / ulp;  // exact (a power of two)
        const float fl = floorf(x);
        if (x - fl == 0.5f || !(x < 8388608.f) || x < 1.f) return false;  // tie / step smaller than an ulp or huge: run the additions
        k_ulp = (int)rintf(x);
        return ((bits + n * k_ulp) >> 23) == (bits >> 23);  // the last sum is still in the binade
    };

    uint32_t count = 0;
    float t0 = near, t1 = t0 + dt, tm = (t0 + t1) * 0.5f;
    bool empty_mode = false, has_pending = false;
    float pending = 0.f;
    while (true) {
        if (!empty_mode) {
            // ---- the next 64 samples of an occupied stretch: lane j = the state after j steps  t0 = t1, t1 = t0 + dt
            float a0, a1, my0, my1;
            int k_ulp;
            if (progression(t1, 64, k_ulp)) {  // t1_j = t1 + j adds of dt = t1's bit pattern + j * k_ulp (see progression())
                const int b1 = __float_as_int(t1);
                my1 = __int_as_float(b1 + lane * k_ulp);
                my0 = lane == 0 ? t0 : __int_as_float(b1 + (lane - 1) * k_ulp);
                a0 = __int_as_float(b1 + 63 * k_ulp);
                a1 = __int_as_float(b1 + 64 * k_ulp);
            } else {
                a0 = t0; a1 = t1; my0 = t0; my1 = t1;
#pragma unroll 1
                for (int j = 0; j < 64; ++j) {  // (rare: a binade boundary inside the window, or a step that ties)
                    if (lane == j) { my0 = a0; my1 = a1; }
                    const float n0 = a1;
                    a1 = n0 + dt;
                    a0 = n0;
                }
            }
            const float mytm = lane == 0 ? tm : (my0 + my1) * 0.5f;  // (lane 0: t_mid as the loop arrived with it)
            float dist;
            const bool occ = visit(mytm, dist);
            const unsigned long long ok = __ballot((mytm < far) && occ);
            const int ff = ok == ~0ull ? 64 : (int)__ffsll((long long)~ok) - 1;  // first sample that ends the stretch
            if (lane < ff) {
                const uint32_t j = count + (uint32_t)lane;
                if (MODE == 1) {
                    t_starts[base + j] = my0;
                    t_ends[base + j] = my1;
                    ray_indices[base + j] = (int64_t)i;
                } else if (MODE == 2) {
                    if (j < cap) row[j] = make_float2(my0, my1);
                }
            }
            count += (uint32_t)ff;
            if (ff == 64) {
                t0 = a0; t1 = a1; tm = (a0 + a1) * 0.5f;
                continue;
            }
            const float ftm = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mytm), ff));
            if (!(ftm < far)) break;
            tm = ftm;  // an EMPTY visit at tm comes next (t0 / t1 are re-derived behind the skip)
            empty_mode = true;
            has_pending = false;
        } else {
            // ---- empty space: the loop visits a subsequence of  b_0 = tm, b_{n+1} = b_n + dt ; lane n holds b_n
            float b = tm, acc = tm;
            int k_ulp;
            const bool fast = progression(tm, 63, k_ulp);
            if (fast) {
                b = __int_as_float(__float_as_int(tm) + lane * k_ulp);
                acc = __int_as_float(__float_as_int(tm) + 63 * k_ulp);
            } else {
#pragma unroll 1
                for (int n = 1; n < 64; ++n) {
                    acc = acc + dt;
                    if (lane == n) b = acc;
                }
            }
            float dist;
            const bool occ = visit(b, dist);
            const float tgt = b + dist;  // t_target of a visit at b: "do t_mid += dt while t_mid < t_target"
            // nxt = smallest m > lane with b_m >= tgt (64: beyond this window), for every lane at once
            int nxt;
            const int d_ulp = __float_as_int(tgt) - __float_as_int(b);  // (same binade: tgt - b in units of its ulp)
            const bool same_binade = ((__float_as_int(tgt) ^ __float_as_int(b)) >> 23) == 0;
            if (fast && __all(same_binade || !(b < far))) {
                // b_m = b + (m - lane) k_ulp ulps exactly, so the first b_m >= tgt is ceil(d_ulp / k_ulp) adds away (at least one:
                // the loop is a do-while)
                int q = (int)(((uint32_t)(d_ulp > 0 ? d_ulp : 0) + (uint32_t)k_ulp - 1u) / (uint32_t)k_ulp);
                q = q < 1 ? 1 : q;
                nxt = (same_binade && q < 64 - lane) ? lane + q : 64;
                if (!same_binade) nxt = 64;  // (only lanes beyond t_max: never reached by the chase)
            } else {
                int lo = lane + 1, hi = 64;
#pragma unroll
                for (int s6 = 0; s6 < 6; ++s6) {
                    const int mid = (lo + hi) >> 1;
                    const float v = __shfl(b, mid < 64 ? mid : 63, 64);
                    if (lo < hi) {
                        if (mid < 64 && !(v < tgt)) hi = mid; else lo = mid + 1;
                    }
                }
                nxt = lo;
            }
            int cur = 0;
            if (has_pending) {  // a skip that began in an earlier window: first b_m (m >= 1) that is not < pending
                const unsigned long long ge = __ballot(!(b < pending)) & ~1ull;
                cur = ge ? (int)__ffsll((long long)ge) - 1 : 64;
            }
            bool done = false;
            const unsigned long long occ_mask = __ballot(occ);
            cur = __builtin_amdgcn_readfirstlane(cur);  // (the walk through the window is wave-uniform: scalar lane reads)
            while (true) {
                if (cur >= 64) {  // the skip runs past this window: go on from b_63 with the same target
                    tm = acc;
                    has_pending = true;
                    break;
                }
                const float bc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), cur));
                if (!(bc < far)) { done = true; break; }
                if ((occ_mask >> cur) & 1ull) {  // an occupied sample: t0 / t1 as the loop derives them behind a skip
                    t0 = bc - half; t1 = bc + half; tm = bc;
                    empty_mode = false;
                    break;
                }
                pending = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tgt), cur));
                cur = __builtin_amdgcn_readlane(nxt, cur);
            }
            if (done) break;
        }
    }
    if (MODE != 1 && lane == 0) num_steps[i] = (int32_t)count;
}

// wave per ray: copy the ray's scratch row to its packed position
__global__ void __launch_bounds__(256)
k_pack_scratch(const float2 *__restrict__ scratch, uint32_t cap, const int32_t *__restrict__ packed_info,
               int64_t *__restrict__ ray_indices, float *__restrict__ t_starts, float *__restrict__ t_ends,
               uint32_t n_rays)
{
    const uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= n_rays) return;
    const uint32_t start = (uint32_t)packed_info[2ull * r], count = (uint32_t)packed_info[2ull * r + 1];
    const float2 *row = scratch + (uint64_t)r * cap;
    for (uint32_t k = lane; k < count; k += 64) {
        const float2 v = row[min(k, cap - 1u)];  // (a count beyond the row is the caller's overflow case: stay in bounds)
        t_starts[start + k] = v.x;
        t_ends[start + k] = v.y;
        ray_indices[start + k] = (int64_t)r;
    }
}

// exclusive scan of per-ray counts by ONE workgroup (n_rays is a few thousand): PACK_BLOCK lanes, each owns a
// contiguous chunk; wave scan + LDS across the waves.  Removes torch.cumsum + stack from the step.
// Two block sizes.  1024 lanes are the fast ones (8,192 counts: 9 us; the lanes' serial chunk loops are latency chains, at
// 256 lanes the same scan takes 15-25 us) -- but a 16-wave workgroup has to wait for a CU with four free wave slots on
// EVERY SIMD, and next to the fp32 MLP kernels of the NeuS steps (one wave per SIMD holding the whole register file) it
// waited for hundreds of microseconds on the marching stream (rocprofv3: 380-430 us "duration" for a 4,096-element scan).
// So: up to 2,048 counts (a NeuS step at the reference's operating point) 256 lanes, more (the NeRF step's 8,192 slots) 1024.
template <int PACK_WAVES>
__device__ __forceinline__ int32_t prefix_total(const int32_t *wave_tot)
{
    int32_t t = 0;
#pragma unroll
    for (int k = 0; k < PACK_WAVES; ++k) t += wave_tot[k];
    return t;
}

constexpr uint32_t PACK_LDS = 16384;  // ray counts up to this are staged through LDS (coalesced loads and stores)
// every lane walks its own contiguous chunk of the staged counts: a chunk of 32 would put all 64 lanes on ONE LDS bank (the
// 256-thread version of this kernel first took 29 us instead of 9) -- one padding word per 32 keeps the lanes on different banks
#define PACK_IDX(k) ((k) + ((k) >> 5))
template <int PACK_BLOCK>
__global__ void __launch_bounds__(PACK_BLOCK)
k_pack_from_counts(const int32_t *__restrict__ counts, int32_t *__restrict__ packed, int32_t *__restrict__ total,
                   uint32_t n, uint32_t capacity, int32_t *__restrict__ stats, const int32_t *__restrict__ n_active)
{
    constexpr int PACK_WAVES = PACK_BLOCK / 64;
    __shared__ int32_t wave_tot[PACK_WAVES];
    extern __shared__ int32_t buf[];  // n words when staged (sized by the launch: a fixed 64 KiB would keep this one-workgroup
                                      // kernel waiting for a CU with that much free LDS next to the step's big kernels)
    const uint32_t tid = threadIdx.x, chunk = (n + PACK_BLOCK - 1) / PACK_BLOCK;
    const uint32_t lo = min(tid * chunk, n), hi = min(lo + chunk, n);
    // slots >= *n_active are dead rays: they were marched (the marching pass runs ahead of the ray count) but keep nothing
    const uint32_t live = n_active ? (uint32_t)max(min(*n_active, (int32_t)n), 0) : n;
    const bool staged = n <= PACK_LDS;
    if (staged) {
        // eight loads in flight per lane: a plain load / LDS-store loop waits for every load (~1 us each) -- at 8,192 rays and
        // 256 lanes that alone was 25 us of this kernel
        for (uint32_t k0 = tid; k0 < n; k0 += PACK_BLOCK * 8) {
            int32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t k = k0 + u * PACK_BLOCK;
                v[u] = (k < n && k < live) ? counts[k] : 0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t k = k0 + u * PACK_BLOCK;
                if (k < n) buf[PACK_IDX(k)] = v[u];
            }
        }
        __syncthreads();
    }
    int32_t s = 0;
    for (uint32_t k = lo; k < hi; ++k) s += staged ? buf[PACK_IDX(k)] : (k < live ? counts[k] : 0);
    // inclusive scan of s across the block
    int32_t v = s;
    const int lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int32_t t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    if (lane == 63) wave_tot[w] = v;
    __syncthreads();
    int32_t prefix = 0;
    for (int k = 0; k < w; ++k) prefix += wave_tot[k];
    int32_t run = prefix + v - s;  // exclusive prefix of this lane's chunk
    for (uint32_t k = lo; k < hi; ++k) {
        int32_t c = staged ? buf[PACK_IDX(k)] : (k < live ? counts[k] : 0), start = run;
        run += c;
        if (capacity) {  // fixed-size sample buffers: rays past the capacity are truncated (and reported)
            start = min(start, (int32_t)capacity);
            c = min(c, (int32_t)capacity - start);
        }
        if (staged) {  // only this lane touches buf[lo, hi): the start goes to its own slot, the count is rebuilt below
            buf[PACK_IDX(k)] = start;
        } else {
            packed[2ull * k] = start;
            packed[2ull * k + 1] = c;
        }
    }
    if (staged) {
        __syncthreads();
        const int32_t end_all = capacity ? min(prefix_total<PACK_WAVES>(wave_tot), (int32_t)capacity) : prefix_total<PACK_WAVES>(wave_tot);
        const bool vec = (reinterpret_cast<uintptr_t>(packed) & 7u) == 0;  // (a caller may hand in a 4-byte aligned view)
        for (uint32_t k = tid; k < n; k += PACK_BLOCK) {  // count = next start - start (truncation included)
            const int32_t a = buf[PACK_IDX(k)], b = k + 1 < n ? buf[PACK_IDX(k + 1)] : end_all;
            if (vec) reinterpret_cast<int2 *>(packed)[k] = make_int2(a, b - a);
            else { packed[2ull * k] = a; packed[2ull * k + 1] = b - a; }
        }
    }
    if (tid == PACK_BLOCK - 1) {
        const int32_t t = prefix + v;
        total[0] = capacity ? min(t, (int32_t)capacity) : t;
        if (stats) {  // int32[6]: [0] last (unclamped) total, [1] largest, [2] #launches truncated, [4..5] u64 running sum
            atomicAdd(reinterpret_cast<unsigned long long *>(stats + 4), (unsigned long long)t);
            atomicMax(&stats[1], t);
            if (capacity && t > (int32_t)capacity) atomicAdd(&stats[2], 1);
            stats[0] = t;
        }
    }
}

__global__ void __launch_bounds__(EW_BLOCK)
k_pack_info(const int64_t *__restrict__ ray_indices, int32_t *__restrict__ packed, uint32_t n, uint32_t n_rays)
{
    const uint32_t r = blockIdx.x * EW_BLOCK + threadIdx.x;
    if (r >= n_rays) return;
    // lower_bound(r) and lower_bound(r+1) on the sorted ray_indices
    uint32_t lo = 0, hi = n;
    while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (ray_indices[m] < (int64_t)r) lo = m + 1; else hi = m; }
    const uint32_t a = lo;
    hi = n;
    while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (ray_indices[m] <= (int64_t)r) lo = m + 1; else hi = m; }
    packed[2ull * r] = (int32_t)a;
    packed[2ull * r + 1] = (int32_t)(lo - a);
}

template <bool INV>
__global__ void __launch_bounds__(EW_BLOCK)
k_contract(const float *__restrict__ x, const float *__restrict__ roi, int type, float *__restrict__ out, uint32_t n)
{
    const uint32_t i = blockIdx.x * EW_BLOCK + threadIdx.x;
    if (i >= n) return;
    Roi r;
#pragma unroll
    for (int k = 0; k < 3; ++k) { r.lo[k] = roi[k]; r.hi[k] = roi[3 + k]; }
    const float p[3] = {x[3ull * i], x[3ull * i + 1], x[3ull * i + 2]};
    float q[3];
    if (INV) apply_contraction_inv(p, r, type, q);
    else apply_contraction(p, r, type, q);
    out[3ull * i] = q[0]; out[3ull * i + 1] = q[1]; out[3ull * i + 2] = q[2];
}

__global__ void __launch_bounds__(EW_BLOCK)
k_grid_query_u8(const float *__restrict__ x, const float *__restrict__ roi, const uint8_t *__restrict__ grid, int3 res,
                int type, uint8_t *__restrict__ out, uint32_t n)
{
    const uint32_t i = blockIdx.x * EW_BLOCK + threadIdx.x;
    if (i >= n) return;
    Roi r;
#pragma unroll
    for (int k = 0; k < 3; ++k) { r.lo[k] = roi[k]; r.hi[k] = roi[3 + k]; }
    const float p[3] = {x[3ull * i], x[3ull * i + 1], x[3ull * i + 2]};
    out[i] = grid_occupied_at(p, r, type, res, grid) ? 1 : 0;
}

// ---- order-preserving stream compaction (the boolean mask after render_visibility) ---------------
// pass 1: per-block kept counts; pass 2 (single block): exclusive scan of block counts; pass 3: scatter.
constexpr int CP_BLOCK = 256;

__device__ __forceinline__ int block_excl_scan(int v, int *lds /*[4]*/, int &block_total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    int prefix = 0;
    for (int k = 0; k < w; ++k) prefix += lds[k];
    block_total = lds[0] + lds[1] + lds[2] + lds[3];
    return prefix + inc - v;
}

__global__ void __launch_bounds__(CP_BLOCK)
k_compact_count(const uint8_t *__restrict__ mask, int32_t *__restrict__ block_counts, uint32_t n)
{
    __shared__ int lds[4];
    const uint32_t i = blockIdx.x * CP_BLOCK + threadIdx.x;
    const int keep = (i < n && mask[i]) ? 1 : 0;
    int tot;
    block_excl_scan(keep, lds, tot);
    if (threadIdx.x == 0) block_counts[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(1024)
k_compact_scan_blocks(int32_t *__restrict__ block_counts, int32_t *__restrict__ n_kept, uint32_t n_blocks)
{
    __shared__ int32_t wave_tot[16];
    const uint32_t tid = threadIdx.x, chunk = (n_blocks + 1023) / 1024;
    const uint32_t lo = min(tid * chunk, n_blocks), hi = min(lo + chunk, n_blocks);
    int32_t s = 0;
    for (uint32_t k = lo; k < hi; ++k) s += block_counts[k];
    int32_t v = s;
    const int lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int32_t t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    if (lane == 63) wave_tot[w] = v;
    __syncthreads();
    int32_t prefix = 0;
    for (int k = 0; k < w; ++k) prefix += wave_tot[k];
    int32_t run = prefix + v - s;
    for (uint32_t k = lo; k < hi; ++k) {
        const int32_t c = block_counts[k];
        block_counts[k] = run;
        run += c;
    }
    if (tid == 1023) n_kept[0] = prefix + v;
}

__global__ void __launch_bounds__(CP_BLOCK)
k_compact_scatter(const uint8_t *__restrict__ mask, const int32_t *__restrict__ block_offsets,
                  const int64_t *__restrict__ ri, const float *__restrict__ t0, const float *__restrict__ t1,
                  int64_t *__restrict__ ri_o, float *__restrict__ t0_o, float *__restrict__ t1_o, uint32_t n)
{
    __shared__ int lds[4];
    const uint32_t i = blockIdx.x * CP_BLOCK + threadIdx.x;
    const int keep = (i < n && mask[i]) ? 1 : 0;
    int tot;
    const int pos = block_excl_scan(keep, lds, tot);
    if (keep) {
        const uint32_t dst = (uint32_t)block_offsets[blockIdx.x] + (uint32_t)pos;
        ri_o[dst] = ri[i];
        t0_o[dst] = t0[i];
        t1_o[dst] = t1[i];
    }
}

}  // namespace

extern "C" int nsr_ray_aabb_intersect(const float *rays_o, const float *rays_d, const float *aabb, float *t_min,
                                      float *t_max, uint32_t n_rays, void *stream)
{
    if (n_rays == 0) return NSR_OK;
    NSR_REQUIRE(rays_o && rays_d && aabb && t_min && t_max, "nsr_ray_aabb_intersect: NULL pointer");
    hipLaunchKernelGGL(k_ray_aabb, dim3(nsr_div_up(n_rays, EW_BLOCK)), dim3(EW_BLOCK), 0, (hipStream_t)stream, rays_o,
                       rays_d, aabb, t_min, t_max, n_rays);
    NSR_CHECK_LAUNCH("nsr_ray_aabb_intersect");
    return NSR_OK;
}

static int check_march(const void *a, const void *b, const void *c, const void *d, const void *e, const void *f,
                       int rx, int ry, int rz, int type, float step)
{
    NSR_REQUIRE(a && b && c && d && e && f, "nsr_ray_march: NULL pointer");
    NSR_REQUIRE(rx > 0 && ry > 0 && rz > 0, "nsr_ray_march: bad grid resolution %d %d %d", rx, ry, rz);
    NSR_REQUIRE(type >= 0 && type <= 2, "nsr_ray_march: bad contraction type %d", type);
    NSR_REQUIRE(step > 0.f, "nsr_ray_march: render_step_size must be > 0");
    return NSR_OK;
}

extern "C" int nsr_ray_march_count(const float *rays_o, const float *rays_d, const float *t_min, const float *t_max,
                                   const float *roi, const uint8_t *grid_binary, int res_x, int res_y, int res_z,
                                   int contraction, float step_size, float cone_angle, int32_t *num_steps,
                                   uint32_t n_rays, void *stream)
{
    if (n_rays == 0) return NSR_OK;
    if (int rc = check_march(rays_o, rays_d, t_min, t_max, roi, grid_binary, res_x, res_y, res_z, contraction, step_size))
        return rc;
    NSR_REQUIRE(num_steps, "nsr_ray_march_count: num_steps is NULL");
    hipLaunchKernelGGL((k_ray_march<false>), dim3(nsr_div_up(n_rays, MARCH_BLOCK)), dim3(MARCH_BLOCK), 0,
                       (hipStream_t)stream, rays_o, rays_d, t_min, t_max, roi, grid_binary,
                       make_int3(res_x, res_y, res_z), contraction, step_size, cone_angle, (const int32_t *)nullptr,
                       num_steps, (int64_t *)nullptr, (float *)nullptr, (float *)nullptr, n_rays);
    NSR_CHECK_LAUNCH("nsr_ray_march_count");
    return NSR_OK;
}

extern "C" int nsr_ray_march_write(const float *rays_o, const float *rays_d, const float *t_min, const float *t_max,
                                   const float *roi, const uint8_t *grid_binary, int res_x, int res_y, int res_z,
                                   int contraction, float step_size, float cone_angle, const int32_t *packed_info,
                                   int64_t *ray_indices, float *t_starts, float *t_ends, uint32_t n_rays, void *stream)
{
    if (n_rays == 0) return NSR_OK;
    if (int rc = check_march(rays_o, rays_d, t_min, t_max, roi, grid_binary, res_x, res_y, res_z, contraction, step_size))
        return rc;
    NSR_REQUIRE(packed_info && ray_indices && t_starts && t_ends, "nsr_ray_march_write: NULL output");
    hipLaunchKernelGGL((k_ray_march<true>), dim3(nsr_div_up(n_rays, MARCH_BLOCK)), dim3(MARCH_BLOCK), 0,
                       (hipStream_t)stream, rays_o, rays_d, t_min, t_max, roi, grid_binary,
                       make_int3(res_x, res_y, res_z), contraction, step_size, cone_angle, packed_info,
                       (int32_t *)nullptr, ray_indices, t_starts, t_ends, n_rays);
    NSR_CHECK_LAUNCH("nsr_ray_march_write");
    return NSR_OK;
}

extern "C" uint64_t nsr_grid_bricks_words64(int res_x, int res_y, int res_z)
{
    if (res_x <= 0 || res_y <= 0 || res_z <= 0 || (res_x & 3) || (res_y & 3) || (res_z & 3)) return 0;
    const uint64_t nb = (uint64_t)(res_x >> 2) * (res_y >> 2) * (res_z >> 2);
    return nb + ((nb + 31) / 32 + 1) / 2;  // bricks (u64 each) followed by the any-bits (u32 words)
}

extern "C" int nsr_grid_pack_bricks(const uint8_t *grid_binary, int res_x, int res_y, int res_z, uint64_t *bricks,
                                    void *stream)
{
    const uint64_t words = nsr_grid_bricks_words64(res_x, res_y, res_z);
    NSR_REQUIRE(words > 0, "nsr_grid_pack_bricks: resolution %dx%dx%d must be positive multiples of 4", res_x, res_y, res_z);
    NSR_REQUIRE(grid_binary && bricks, "nsr_grid_pack_bricks: NULL pointer");
    const uint32_t nb = (uint32_t)((res_x >> 2) * (res_y >> 2) * (res_z >> 2));
    NSR_REQUIRE((nb & 63u) == 0, "nsr_grid_pack_bricks: brick count must be a multiple of 64");
    hipLaunchKernelGGL(k_pack_bricks, dim3(nsr_div_up(nb, 256)), dim3(256), 0, (hipStream_t)stream, grid_binary,
                       make_int3(res_x, res_y, res_z), (unsigned long long *)bricks, (uint32_t *)(bricks + nb), nb);
    NSR_CHECK_LAUNCH("nsr_grid_pack_bricks");
    return NSR_OK;
}

extern "C" uint32_t nsr_ray_march_capacity(const float *roi_host, float step_size)
{
    // samples are only emitted inside the roi (AABB type): at most diag/step of them, +3 for rounding slack
    const float dx = roi_host[3] - roi_host[0], dy = roi_host[4] - roi_host[1], dz = roi_host[5] - roi_host[2];
    const double diag = sqrt((double)dx * dx + (double)dy * dy + (double)dz * dz);
    const double n = diag / (double)step_size + 3.0;
    return n > 65536.0 ? 0u : (uint32_t)n;
}

static int g_march_wave = 1;  // wave-per-ray kernel: 0 never, 1 for launches of <= 32,768 rays (default), 2 always
extern "C" int nsr_ray_march_wave_mode(int mode)
{
    const int old = g_march_wave;
    g_march_wave = mode < 0 ? 0 : (mode > 2 ? 2 : mode);
    return old;
}

static uint32_t g_rays_per_wave = 0;  // 0: by the ray count
extern "C" uint32_t nsr_ray_march_rays_per_wave(uint32_t rays_per_wave)
{
    const uint32_t old = g_rays_per_wave;
    g_rays_per_wave = rays_per_wave > 64u ? 64u : rays_per_wave;
    return old;
}

static int launch_bricks(int mode, const float *rays_o, const float *rays_d, const float *t_min, const float *t_max,
                         const float *roi, const uint64_t *bricks, int rx, int ry, int rz, int type, float step,
                         float cone, const int32_t *packed, int32_t *num_steps, int64_t *ri, float *t0, float *t1,
                         float *scratch, uint32_t cap, uint32_t n_rays, void *stream)
{
    const uint32_t nb = (uint32_t)((rx >> 2) * (ry >> 2) * (rz >> 2));
    const uint32_t n_words = (nb + 31) / 32;
    const uint32_t *any_bits = (const uint32_t *)(bricks + nb);
    const size_t lds = n_words * sizeof(uint32_t);
    NSR_REQUIRE(lds <= 64 * 1024, "nsr_ray_march(bricks): grid too large for the LDS any-bit table");
    // wave-per-ray kernel: bounded scenes (AABB, fixed step) and launches small enough that the chip is not full of rays anyway
    if (type == NSR_CONTRACT_AABB && cone == 0.f && g_march_wave != 0 && (g_march_wave == 2 || n_rays <= 32768u)) {
        const dim3 wgrid(nsr_div_up((uint64_t)n_rays * 64ull, MARCH_WAVE_BLOCK)), wblock(MARCH_WAVE_BLOCK);
        const int3 wres = make_int3(rx, ry, rz);
        const bool wpow2 = !(rx & (rx - 1)) && !(ry & (ry - 1)) && !(rz & (rz - 1));
#define NSR_LAUNCH_WAVE(M, P)                                                                                          \
    hipLaunchKernelGGL((k_ray_march_wave<M, P>), wgrid, wblock, lds, (hipStream_t)stream, rays_o, rays_d, t_min, t_max, \
                       roi, (const unsigned long long *)bricks, any_bits, n_words, wres, step, packed, num_steps, ri, t0, \
                       t1, (float2 *)scratch, cap, n_rays)
        if (wpow2) {
            if (mode == 0) NSR_LAUNCH_WAVE(0, true);
            else if (mode == 1) NSR_LAUNCH_WAVE(1, true);
            else NSR_LAUNCH_WAVE(2, true);
        } else {
            if (mode == 0) NSR_LAUNCH_WAVE(0, false);
            else if (mode == 1) NSR_LAUNCH_WAVE(1, false);
            else NSR_LAUNCH_WAVE(2, false);
        }
#undef NSR_LAUNCH_WAVE
        return NSR_OK;
    }
    // rays per wave: full waves once the launch fills the chip's SIMDs several times over (the window-batched launches of
    // the asynchronous trainer), 16 or 8 lanes per wave for the single ray sets of the model-interface path
    // (nsr_ray_march_rays_per_wave overrides: A/B)
    uint32_t rpw = g_rays_per_wave ? g_rays_per_wave : (n_rays >= 65536u ? 64u : (n_rays >= 16384u ? 16u : 8u));
    const dim3 grid(nsr_div_up(nsr_div_up(n_rays, rpw) * 64ull, MARCH_BLOCK)), block(MARCH_BLOCK);
    const int3 res = make_int3(rx, ry, rz);
    const bool pow2 = !(rx & (rx - 1)) && !(ry & (ry - 1)) && !(rz & (rz - 1));
#define NSR_LAUNCH_BRICKS(M, P)                                                                                        \
    hipLaunchKernelGGL((k_ray_march_bricks<M, P>), grid, block, lds, (hipStream_t)stream, rays_o, rays_d, t_min, t_max, \
                       roi, (const unsigned long long *)bricks, any_bits, n_words, res, type, step, cone, packed,      \
                       num_steps, ri, t0, t1, (float2 *)scratch, cap, n_rays, rpw)
    if (pow2) {
        if (mode == 0) NSR_LAUNCH_BRICKS(0, true);
        else if (mode == 1) NSR_LAUNCH_BRICKS(1, true);
        else NSR_LAUNCH_BRICKS(2, true);
    } else {
        if (mode == 0) NSR_LAUNCH_BRICKS(0, false);
        else if (mode == 1) NSR_LAUNCH_BRICKS(1, false);
        else NSR_LAUNCH_BRICKS(2, false);
    }
#undef NSR_LAUNCH_BRICKS
    return NSR_OK;
}

extern "C" int nsr_ray_march_bricks_count(const float *rays_o, const float *rays_d, const float *t_min,
                                          const float *t_max, const float *roi, const uint64_t *bricks, int res_x,
                                          int res_y, int res_z, int contraction, float step_size, float cone_angle,
                                          int32_t *num_steps, float *scratch, uint32_t capacity, uint32_t n_rays,
                                          void *stream)
{
    if (n_rays == 0) return NSR_OK;
    if (int rc = check_march(rays_o, rays_d, t_min, t_max, roi, bricks, res_x, res_y, res_z, contraction, step_size))
        return rc;
    NSR_REQUIRE(num_steps, "nsr_ray_march_bricks_count: num_steps is NULL");
    NSR_REQUIRE((scratch == nullptr) == (capacity == 0), "nsr_ray_march_bricks_count: scratch and capacity go together");
    if (int rc = launch_bricks(scratch ? 2 : 0, rays_o, rays_d, t_min, t_max, roi, bricks, res_x, res_y, res_z,
                               contraction, step_size, cone_angle, nullptr, num_steps, nullptr, nullptr, nullptr,
                               scratch, capacity, n_rays, stream))
        return rc;
    NSR_CHECK_LAUNCH("nsr_ray_march_bricks_count");
    return NSR_OK;
}

extern "C" int nsr_ray_march_bricks_write(const float *rays_o, const float *rays_d, const float *t_min,
                                          const float *t_max, const float *roi, const uint64_t *bricks, int res_x,
                                          int res_y, int res_z, int contraction, float step_size, float cone_angle,
                                          const int32_t *packed_info, const float *scratch, uint32_t capacity,
                                          int64_t *ray_indices, float *t_starts, float *t_ends, uint32_t n_rays,
                                          void *stream)
{
    if (n_rays == 0) return NSR_OK;
    NSR_REQUIRE(packed_info && ray_indices && t_starts && t_ends, "nsr_ray_march_bricks_write: NULL output");
    if (scratch) {
        hipLaunchKernelGGL(k_pack_scratch, dim3(nsr_div_up(n_rays, 4)), dim3(256), 0, (hipStream_t)stream,
                           (const float2 *)scratch, capacity, packed_info, ray_indices, t_starts, t_ends, n_rays);
    } else {
        if (int rc = check_march(rays_o, rays_d, t_min, t_max, roi, bricks, res_x, res_y, res_z, contraction, step_size))
            return rc;
        if (int rc = launch_bricks(1, rays_o, rays_d, t_min, t_max, roi, bricks, res_x, res_y, res_z, contraction,
                                   step_size, cone_angle, packed_info, nullptr, ray_indices, t_starts, t_ends, nullptr, 0,
                                   n_rays, stream))
            return rc;
    }
    NSR_CHECK_LAUNCH("nsr_ray_march_bricks_write");
    return NSR_OK;
}

extern "C" int nsr_pack_from_counts_capped(const int32_t *num_steps, int32_t *packed_info, int32_t *total,
                                           uint32_t n_rays, uint32_t capacity, int32_t *stats, const int32_t *n_active,
                                           void *stream)
{
    NSR_REQUIRE(total, "nsr_pack_from_counts: total is NULL");
    NSR_REQUIRE(n_rays == 0 || (num_steps && packed_info), "nsr_pack_from_counts: NULL pointer");
    NSR_REQUIRE(capacity < 0x7fffffffu, "nsr_pack_from_counts: capacity must fit int32");
    NSR_REQUIRE(!stats || ((uintptr_t)stats & 7u) == 0, "nsr_pack_from_counts: stats must be 8-byte aligned");
    const size_t lds = n_rays <= PACK_LDS ? (size_t)(n_rays + (n_rays >> 5) + 1) * sizeof(int32_t) : 0;
    if (n_rays > 2048)
        hipLaunchKernelGGL(k_pack_from_counts<1024>, dim3(1), dim3(1024), lds, (hipStream_t)stream, num_steps, packed_info,
                           total, n_rays, capacity, stats, n_active);
    else
        hipLaunchKernelGGL(k_pack_from_counts<256>, dim3(1), dim3(256), lds, (hipStream_t)stream, num_steps, packed_info,
                           total, n_rays, capacity, stats, n_active);
    NSR_CHECK_LAUNCH("nsr_pack_from_counts");
    return NSR_OK;
}

extern "C" int nsr_pack_from_counts(const int32_t *num_steps, int32_t *packed_info, int32_t *total, uint32_t n_rays,
                                    void *stream)
{
    return nsr_pack_from_counts_capped(num_steps, packed_info, total, n_rays, 0, nullptr, nullptr, stream);
}

extern "C" int nsr_pack_info(const int64_t *ray_indices, int32_t *packed_info, uint32_t n, uint32_t n_rays,
                             void *stream)
{
    if (n_rays == 0) return NSR_OK;
    NSR_REQUIRE(packed_info && (n == 0 || ray_indices), "nsr_pack_info: NULL pointer");
    hipLaunchKernelGGL(k_pack_info, dim3(nsr_div_up(n_rays, EW_BLOCK)), dim3(EW_BLOCK), 0, (hipStream_t)stream,
                       ray_indices, packed_info, n, n_rays);
    NSR_CHECK_LAUNCH("nsr_pack_info");
    return NSR_OK;
}

extern "C" int nsr_contract(const float *x, const float *roi, int contraction, float *out, uint32_t n, void *stream)
{
    if (n == 0) return NSR_OK;
    NSR_REQUIRE(x && roi && out, "nsr_contract: NULL pointer");
    hipLaunchKernelGGL((k_contract<false>), dim3(nsr_div_up(n, EW_BLOCK)), dim3(EW_BLOCK), 0, (hipStream_t)stream, x,
                       roi, contraction, out, n);
    NSR_CHECK_LAUNCH("nsr_contract");
    return NSR_OK;
}

extern "C" int nsr_contract_inv(const float *x, const float *roi, int contraction, float *out, uint32_t n,
                                void *stream)
{
    if (n == 0) return NSR_OK;
    NSR_REQUIRE(x && roi && out, "nsr_contract_inv: NULL pointer");
    hipLaunchKernelGGL((k_contract<true>), dim3(nsr_div_up(n, EW_BLOCK)), dim3(EW_BLOCK), 0, (hipStream_t)stream, x, roi,
                       contraction, out, n);
    NSR_CHECK_LAUNCH("nsr_contract_inv");
    return NSR_OK;
}

extern "C" int nsr_grid_query_u8(const float *x, const float *roi, const uint8_t *grid, int res_x, int res_y,
                                 int res_z, int contraction, uint8_t *out, uint32_t n, void *stream)
{
    if (n == 0) return NSR_OK;
    NSR_REQUIRE(x && roi && grid && out, "nsr_grid_query_u8: NULL pointer");
    hipLaunchKernelGGL(k_grid_query_u8, dim3(nsr_div_up(n, EW_BLOCK)), dim3(EW_BLOCK), 0, (hipStream_t)stream, x, roi,
                       grid, make_int3(res_x, res_y, res_z), contraction, out, n);
    NSR_CHECK_LAUNCH("nsr_grid_query_u8");
    return NSR_OK;
}

extern "C" int nsr_compact_samples(const uint8_t *mask, const int64_t *ray_indices, const float *t_starts,
                                   const float *t_ends, int64_t *ray_indices_out, float *t_starts_out,
                                   float *t_ends_out, int32_t *n_kept, int32_t *block_scratch, uint32_t n,
                                   void *stream)
{
    NSR_REQUIRE(n_kept && block_scratch, "nsr_compact_samples: NULL scratch");
    const uint32_t nb = nsr_div_up(n, CP_BLOCK);
    if (n > 0) {
        NSR_REQUIRE(mask && ray_indices && t_starts && t_ends && ray_indices_out && t_starts_out && t_ends_out,
                    "nsr_compact_samples: NULL pointer");
        hipLaunchKernelGGL(k_compact_count, dim3(nb), dim3(CP_BLOCK), 0, (hipStream_t)stream, mask, block_scratch, n);
    }
    hipLaunchKernelGGL(k_compact_scan_blocks, dim3(1), dim3(1024), 0, (hipStream_t)stream, block_scratch, n_kept, nb);
    if (n > 0)
        hipLaunchKernelGGL(k_compact_scatter, dim3(nb), dim3(CP_BLOCK), 0, (hipStream_t)stream, mask, block_scratch,
                           ray_indices, t_starts, t_ends, ray_indices_out, t_starts_out, t_ends_out, n);
    NSR_CHECK_LAUNCH("nsr_compact_samples");
    return NSR_OK;
}
