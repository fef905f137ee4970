/* Oracle (CPU, scalar fp32) restatement of the nerfacc==0.3.3 kernels the reference calls.
 * TEST INFRASTRUCTURE ONLY -- never linked into or called from the product path.
 *
 * nerfacc 0.3.3 (reference requirements.txt:3) is a third-party dependency that is not under
 * /root/reference; this file restates its published algorithm.
 * Reference call sites each function stands behind:
 *   nsro_ray_aabb_intersect        models/neus.py:153 ; inside ray_marching for models/nerf.py:85, neus.py:212
 *   nsro_ray_march                 models/nerf.py:83 ; models/neus.py:159,210
 *   nsro_contract / _contract_inv  OccupancyGrid._update behind models/nerf.py:55, neus.py:109-111
 *   nsro_grid_query                nerfacc query_grid (OccupancyGrid.query_occ)
 *   nsro_transmittance_from_sigma  render_weight_from_density  models/nerf.py:105 ; neus.py:181
 *   nsro_transmittance_from_alpha  render_weight_from_alpha    models/neus.py:237 ; render_visibility
 *
 * Arithmetic contract (what the HIP kernels must reproduce BIT-EXACTLY for t_starts/t_ends/indices):
 *   - every operation is a separately rounded IEEE fp32 op (build with -ffp-contract=off), except the
 *     sample position  p = fmaf(t, d, o)  which is one fused op (nvcc contracts that expression);
 *   - division and sqrt are correctly rounded; no flush-to-zero;
 *   - two-pass marching: pass 1 counts, host prefix-sum, pass 2 writes (same loop both times).
 * PARITY STATUS: unpinned (no nerfacc binary or vectors available); KATs in tests/test_oracle_kat.py.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

enum { NSR_AABB = 0, NSR_UN_BOUNDED_TANH = 1, NSR_UN_BOUNDED_SPHERE = 2 };

static inline float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
static inline float signf(float v) { return (float)((v > 0.0f) - (v < 0.0f)); }

/* ---- slab test.  miss => 1e10/1e10 ; hit => (max(tmin,0), tmax) ------------------------------ */
static void aabb_one(const float *o, const float *d, const float *aabb, float *near, float *far)
{
    float tmin = (aabb[0] - o[0]) / d[0], tmax = (aabb[3] - o[0]) / d[0];
    if (tmin > tmax) { float s = tmin; tmin = tmax; tmax = s; }
    float tymin = (aabb[1] - o[1]) / d[1], tymax = (aabb[4] - o[1]) / d[1];
    if (tymin > tymax) { float s = tymin; tymin = tymax; tymax = s; }
    if (tmin > tymax || tymin > tmax) { *near = 1e10f; *far = 1e10f; return; }
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = (aabb[2] - o[2]) / d[2], tzmax = (aabb[5] - o[2]) / d[2];
    if (tzmin > tzmax) { float s = tzmin; tzmin = tzmax; tzmax = s; }
    if (tmin > tzmax || tzmin > tmax) { *near = 1e10f; *far = 1e10f; return; }
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    *near = tmin > 0.0f ? tmin : 0.0f; /* the kernel wrapper clamps t_min to >= 0; t_max is left as is */
    *far = tmax;
}

void nsro_ray_aabb_intersect(int64_t n, const float *rays_o, const float *rays_d, const float *aabb,
                             float *t_min, float *t_max)
{
    for (int64_t i = 0; i < n; ++i)
        aabb_one(rays_o + 3 * i, rays_d + 3 * i, aabb, t_min + i, t_max + i);
}

/* ---- contraction -------------------------------------------------------------- */
static void roi_to_unit(const float *p, const float *roi, float *u)
{
    for (int k = 0; k < 3; ++k) u[k] = (p[k] - roi[k]) / (roi[3 + k] - roi[k]);
}

static void apply_contraction(const float *p, const float *roi, int type, float *u)
{
    roi_to_unit(p, roi, u);
    if (type == NSR_UN_BOUNDED_SPHERE) {
        float v[3];
        for (int k = 0; k < 3; ++k) v[k] = u[k] * 2.0f - 1.0f;
        float n = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        if (n > 1.0f) {
            float s = 2.0f - 1.0f / n;
            for (int k = 0; k < 3; ++k) v[k] = s * (v[k] / n);
        }
        for (int k = 0; k < 3; ++k) u[k] = v[k] * 0.25f + 0.5f;
    } else if (type == NSR_UN_BOUNDED_TANH) {
        /* roi -> [0.25,0.75]^3 : tanh(2*atanh(0.5) * (u-0.5)) / 2 + 0.5 */
        for (int k = 0; k < 3; ++k) u[k] = tanhf((u[k] - 0.5f) * 1.0986122886681098f) * 0.5f + 0.5f;
    }
}

static void apply_contraction_inv(const float *u, const float *roi, int type, float *p)
{
    float w[3] = {u[0], u[1], u[2]};
    if (type == NSR_UN_BOUNDED_SPHERE) {
        float v[3];
        for (int k = 0; k < 3; ++k) v[k] = (u[k] - 0.5f) * 4.0f;
        float n = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        if (n > 1.0f) {
            float s = 1.0f / (2.0f - n);
            for (int k = 0; k < 3; ++k) v[k] = (v[k] / n) * s;
        }
        for (int k = 0; k < 3; ++k) w[k] = v[k] * 0.5f + 0.5f;
    } else if (type == NSR_UN_BOUNDED_TANH) {
        for (int k = 0; k < 3; ++k) {
            float t = clampf((u[k] - 0.5f) * 2.0f, -1.0f + 1e-6f, 1.0f - 1e-6f);
            w[k] = atanhf(t) / 1.0986122886681098f + 0.5f;
        }
    }
    for (int k = 0; k < 3; ++k) p[k] = w[k] * (roi[3 + k] - roi[k]) + roi[k];
}

void nsro_contract(int64_t n, const float *x, const float *roi, int type, float *out)
{
    for (int64_t i = 0; i < n; ++i) apply_contraction(x + 3 * i, roi, type, out + 3 * i);
}

void nsro_contract_inv(int64_t n, const float *x, const float *roi, int type, float *out)
{
    for (int64_t i = 0; i < n; ++i) apply_contraction_inv(x + 3 * i, roi, type, out + 3 * i);
}

static int grid_idx_at(const float *u, const int *res)
{
    int ix = (int)(u[0] * (float)res[0]), iy = (int)(u[1] * (float)res[1]), iz = (int)(u[2] * (float)res[2]);
    ix = ix < 0 ? 0 : (ix > res[0] - 1 ? res[0] - 1 : ix);
    iy = iy < 0 ? 0 : (iy > res[1] - 1 ? res[1] - 1 : iy);
    iz = iz < 0 ? 0 : (iz > res[2] - 1 ? res[2] - 1 : iz);
    return ix * res[1] * res[2] + iy * res[2] + iz;
}

static int grid_occupied_at(const float *p, const float *roi, int type, const int *res, const uint8_t *grid)
{
    if (type == NSR_AABB && (p[0] < roi[0] || p[0] > roi[3] || p[1] < roi[1] || p[1] > roi[4] ||
                             p[2] < roi[2] || p[2] > roi[5]))
        return 0;
    float u[3];
    apply_contraction(p, roi, type, u);
    return grid[grid_idx_at(u, res)] != 0;
}

/* grid_query on float occupancies / bools: out-of-roi (AABB) => 0 */
void nsro_grid_query_f32(int64_t n, const float *x, const float *roi, const int *res, const float *grid,
                         int type, float *out)
{
    for (int64_t i = 0; i < n; ++i) {
        const float *p = x + 3 * i;
        if (type == NSR_AABB && (p[0] < roi[0] || p[0] > roi[3] || p[1] < roi[1] || p[1] > roi[4] ||
                                 p[2] < roi[2] || p[2] > roi[5])) { out[i] = 0.0f; continue; }
        float u[3];
        apply_contraction(p, roi, type, u);
        out[i] = grid[grid_idx_at(u, res)];
    }
}

void nsro_grid_query_u8(int64_t n, const float *x, const float *roi, const int *res, const uint8_t *grid,
                        int type, uint8_t *out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = (uint8_t)grid_occupied_at(x + 3 * i, roi, type, res, grid);
}

/* ---- ray marching -------------------------------------------------------------- */
static inline float calc_dt(float t, float cone_angle, float dt_min, float dt_max)
{
    return clampf(t * cone_angle, dt_min, dt_max);
}

static float distance_to_next_voxel(const float *p, const float *d, const float *inv_d, const float *roi,
                                    const int *res)
{
    float u[3], t = 0.0f;
    roi_to_unit(p, roi, u);
    for (int k = 0; k < 3; ++k) {
        float r = (float)res[k];
        float x = u[k] * r;
        float tk = ((floorf(x + 0.5f + 0.5f * signf(d[k])) - x) * inv_d[k]) / r * (roi[3 + k] - roi[k]);
        t = (k == 0) ? tk : fminf(t, tk);
    }
    return fmaxf(t, 0.0f);
}

/* One ray.  When t_starts==NULL only counts.  Returns the number of samples. */
static int march_one(const float *o, const float *d, float near, float far, const float *roi, const int *res,
                     const uint8_t *grid, int type, float step, float cone_angle, int64_t ray_id,
                     int64_t *ray_indices, float *t_starts, float *t_ends)
{
    const float inv_d[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
    const float dt_min = step, dt_max = 1e10f;
    int j = 0;
    float t0 = near;
    float dt = calc_dt(t0, cone_angle, dt_min, dt_max);
    float t1 = t0 + dt;
    float t_mid = (t0 + t1) * 0.5f;
    while (t_mid < far) {
        const float p[3] = {fmaf(t_mid, d[0], o[0]), fmaf(t_mid, d[1], o[1]), fmaf(t_mid, d[2], o[2])};
        if (grid_occupied_at(p, roi, type, res, grid)) {
            if (t_starts) { t_starts[j] = t0; t_ends[j] = t1; ray_indices[j] = ray_id; }
            ++j;
            t0 = t1;
            t1 = t0 + calc_dt(t0, cone_angle, dt_min, dt_max);
            t_mid = (t0 + t1) * 0.5f;
        } else if (type == NSR_AABB) {
            float t_target = t_mid + distance_to_next_voxel(p, d, inv_d, roi, res);
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
    return j;
}

/* pass 1: packed_info == NULL -> fills num_steps[n_rays]
 * pass 2: packed_info[n_rays,2] = (start,count) -> fills ray_indices/t_starts/t_ends */
void nsro_ray_march(int64_t n_rays, const float *rays_o, const float *rays_d, const float *t_min,
                    const float *t_max, const float *roi, const int *res, const uint8_t *grid, int type,
                    float step, float cone_angle, const int32_t *packed_info, int32_t *num_steps,
                    int64_t *ray_indices, float *t_starts, float *t_ends)
{
    for (int64_t i = 0; i < n_rays; ++i) {
        if (!packed_info) {
            num_steps[i] = march_one(rays_o + 3 * i, rays_d + 3 * i, t_min[i], t_max[i], roi, res, grid, type,
                                     step, cone_angle, i, NULL, NULL, NULL);
        } else {
            int64_t base = packed_info[2 * i];
            march_one(rays_o + 3 * i, rays_d + 3 * i, t_min[i], t_max[i], roi, res, grid, type, step,
                      cone_angle, i, ray_indices + base, t_starts + base, t_ends + base);
        }
    }
}

/* ---- sequential segmented transmittance (the "naive" per-ray path of nerfacc) ------------------ */
/* T_i = exp(-sum_{j<i in ray} sigma_j*dt_j)   (ray_indices sorted, contiguous segments) */
void nsro_transmittance_from_sigma(int64_t n, const int64_t *ray_indices, const float *t_starts,
                                   const float *t_ends, const float *sigmas, float *trans)
{
    float cum = 0.0f;
    for (int64_t i = 0; i < n; ++i) {
        if (i == 0 || ray_indices[i] != ray_indices[i - 1]) cum = 0.0f;
        trans[i] = expf(-cum);
        cum += sigmas[i] * (t_ends[i] - t_starts[i]);
    }
}

/* T_i = prod_{j<i in ray} (1-alpha_j) */
void nsro_transmittance_from_alpha(int64_t n, const int64_t *ray_indices, const float *alphas, float *trans)
{
    float T = 1.0f;
    for (int64_t i = 0; i < n; ++i) {
        if (i == 0 || ray_indices[i] != ray_indices[i - 1]) T = 1.0f;
        trans[i] = T;
        T *= (1.0f - alphas[i]);
    }
}
