"""Oracle restatement of the nerfacc==0.3.3 surface the reference calls.  TEST INFRASTRUCTURE ONLY.

Mirrors the names / argument meaning of nerfacc 0.3.3 (reference ``requirements.txt:3``) so that the
reference's ``models/nerf.py`` / ``models/neus.py`` / ``models/geometry.py`` import it unchanged:

    ContractionType, OccupancyGrid, ray_marching, render_weight_from_density,
    render_weight_from_alpha, accumulate_along_rays, intersection.ray_aabb_intersect
    (reference call sites: models/nerf.py:11,37,55,83,105-108 ; models/neus.py:11-12,64,70,109,111,
     153,159,181-184,210,237-242 ; models/geometry.py:14)

Sequential fp32 pieces (slab test, marcher, contraction) run in ``oracle/csrc/nerfacc_ref.c``; the
segmented scans are evaluated here in fp64 (a *more* accurate checker than nerfacc's fp32 scan) with
nerfacc's backward formulas restated as custom autograd Functions.
PARITY STATUS: unpinned (see oracle/__init__.py).
"""
import ctypes
import enum

import numpy as np
import torch

from .. import build as _build

_lib = ctypes.CDLL(_build.build())
_f = ctypes.POINTER(ctypes.c_float)
_i32 = ctypes.POINTER(ctypes.c_int32)
_i64 = ctypes.POINTER(ctypes.c_int64)
_u8 = ctypes.POINTER(ctypes.c_uint8)


def _p(t, ty):
    return None if t is None else ctypes.cast(t.data_ptr(), ty)


def _f32c(t):
    return t.detach().to(torch.float32).contiguous().cpu()


class ContractionType(enum.Enum):
    AABB = 0
    UN_BOUNDED_TANH = 1
    UN_BOUNDED_SPHERE = 2

    def to_cpp_version(self):
        return self.value


# ------------------------------------------------------------------------------------------------
# intersection / contraction / grid query
# ------------------------------------------------------------------------------------------------
@torch.no_grad()
def ray_aabb_intersect(rays_o, rays_d, aabb):
    o, d, a = _f32c(rays_o), _f32c(rays_d), _f32c(aabb)
    n = o.shape[0]
    t_min, t_max = torch.empty(n), torch.empty(n)
    _lib.nsro_ray_aabb_intersect(ctypes.c_int64(n), _p(o, _f), _p(d, _f), _p(a, _f), _p(t_min, _f), _p(t_max, _f))
    return t_min, t_max


@torch.no_grad()
def contract(x, roi, type=ContractionType.AABB):
    xx, r = _f32c(x), _f32c(roi)
    out = torch.empty_like(xx)
    _lib.nsro_contract(ctypes.c_int64(xx.shape[0]), _p(xx, _f), _p(r, _f), ctypes.c_int(type.value), _p(out, _f))
    return out


@torch.no_grad()
def contract_inv(x, roi, type=ContractionType.AABB):
    xx, r = _f32c(x), _f32c(roi)
    out = torch.empty_like(xx)
    _lib.nsro_contract_inv(ctypes.c_int64(xx.shape[0]), _p(xx, _f), _p(r, _f), ctypes.c_int(type.value), _p(out, _f))
    return out


@torch.no_grad()
def query_grid(samples, grid_roi, grid_values, grid_type):
    xx, r = _f32c(samples), _f32c(grid_roi)
    res = torch.tensor(list(grid_values.shape), dtype=torch.int32)
    n = xx.shape[0]
    if grid_values.dtype == torch.bool:
        g = grid_values.contiguous().view(torch.uint8)
        out = torch.empty(n, dtype=torch.uint8)
        _lib.nsro_grid_query_u8(ctypes.c_int64(n), _p(xx, _f), _p(r, _f), _p(res, _i32), _p(g, _u8),
                                ctypes.c_int(grid_type.value), _p(out, _u8))
        return out.bool()
    g = _f32c(grid_values)
    out = torch.empty(n)
    _lib.nsro_grid_query_f32(ctypes.c_int64(n), _p(xx, _f), _p(r, _f), _p(res, _i32), _p(g, _f),
                             ctypes.c_int(grid_type.value), _p(out, _f))
    return out


# ------------------------------------------------------------------------------------------------
# occupancy grid (nn.Module: its buffers ride in the reference's state_dict, models/nerf.py:37)
# ------------------------------------------------------------------------------------------------
def _meshgrid3d(res):
    return torch.stack(torch.meshgrid(
        [torch.arange(res[0]), torch.arange(res[1]), torch.arange(res[2])], indexing="ij"), dim=-1).long()


class OccupancyGrid(torch.nn.Module):
    NUM_DIM = 3

    def __init__(self, roi_aabb, resolution=128, contraction_type=ContractionType.AABB):
        super().__init__()
        if isinstance(resolution, int):
            resolution = [resolution] * self.NUM_DIM
        if isinstance(resolution, (list, tuple)):
            resolution = torch.tensor(resolution, dtype=torch.int32)
        if isinstance(roi_aabb, (list, tuple)):
            roi_aabb = torch.tensor(roi_aabb, dtype=torch.float32)
        assert resolution.shape == (3,) and roi_aabb.shape == (6,)
        self.num_cells = int(resolution.prod().item())
        self.register_buffer("_roi_aabb", roi_aabb.clone().float())
        self.register_buffer("_binary", torch.zeros(resolution.tolist(), dtype=torch.bool))
        self._contraction_type = contraction_type
        self.register_buffer("resolution", resolution)
        self.register_buffer("occs", torch.zeros(self.num_cells))
        self.register_buffer("grid_coords", _meshgrid3d(resolution.tolist()).reshape(self.num_cells, 3),
                             persistent=False)
        self.register_buffer("grid_indices", torch.arange(self.num_cells), persistent=False)

    roi_aabb = property(lambda self: self._roi_aabb)
    binary = property(lambda self: self._binary)
    contraction_type = property(lambda self: self._contraction_type)
    device = property(lambda self: self._roi_aabb.device)

    @torch.no_grad()
    def _sample_uniform_and_occupied_cells(self, n):
        uniform = torch.randint(self.num_cells, (n,))
        occupied = torch.nonzero(self._binary.flatten())[:, 0]
        if n < len(occupied):
            occupied = occupied[torch.randint(len(occupied), (n,))]
        return torch.cat([uniform, occupied], dim=0)

    @torch.no_grad()
    def _update_cells(self, indices, jitter, occ_eval_fn, occ_thre=0.01, ema_decay=0.95):
        """Deterministic part of ``_update``: jitter in [0,1)^3 per selected cell."""
        x = (self.grid_coords[indices] + jitter) / self.resolution
        if self._contraction_type == ContractionType.UN_BOUNDED_SPHERE:
            mask = (x - 0.5).norm(dim=1) < 0.5
            x, indices = x[mask], indices[mask]
        x = contract_inv(x, self._roi_aabb, self._contraction_type)
        occ = occ_eval_fn(x).squeeze(-1)
        self.occs[indices] = torch.maximum(self.occs[indices] * ema_decay, occ)
        self._binary = (self.occs > torch.clamp(self.occs.mean(), max=occ_thre)).view(self._binary.shape)

    @torch.no_grad()
    def _update(self, step, occ_eval_fn, occ_thre=0.01, ema_decay=0.95, warmup_steps=256):
        indices = self.grid_indices if step < warmup_steps else \
            self._sample_uniform_and_occupied_cells(self.num_cells // 4)
        jitter = torch.rand(indices.shape[0], 3)
        self._update_cells(indices, jitter, occ_eval_fn, occ_thre, ema_decay)

    @torch.no_grad()
    def every_n_step(self, step, occ_eval_fn, occ_thre=1e-2, ema_decay=0.95, warmup_steps=256, n=16):
        if not self.training:
            raise RuntimeError("You should only call this function only during training. "
                               "Please call _update() directly if you want to update the field during inference.")
        if step % n == 0 and self.training:
            self._update(step=step, occ_eval_fn=occ_eval_fn, occ_thre=occ_thre, ema_decay=ema_decay,
                         warmup_steps=warmup_steps)

    @torch.no_grad()
    def query_occ(self, samples):
        return query_grid(samples, self._roi_aabb, self.binary, self.contraction_type)


# ------------------------------------------------------------------------------------------------
# packing helpers
# ------------------------------------------------------------------------------------------------
@torch.no_grad()
def pack_info(ray_indices, n_rays=None):
    if n_rays is None:
        n_rays = int(ray_indices.max()) + 1 if ray_indices.numel() else 0
    cnt = torch.bincount(ray_indices.long(), minlength=n_rays)
    start = torch.cumsum(cnt, 0) - cnt
    return torch.stack([start, cnt], dim=-1).int()


@torch.no_grad()
def unpack_info(packed_info, n_samples=None):
    cnt = packed_info[:, 1].long()
    return torch.repeat_interleave(torch.arange(packed_info.shape[0]), cnt)


# ------------------------------------------------------------------------------------------------
# segmented scans (fp64 evaluation, nerfacc's backward formulas)
# ------------------------------------------------------------------------------------------------
def _seg_start_index(ray_indices):
    n = ray_indices.shape[0]
    is_start = torch.ones(n, dtype=torch.bool)
    is_start[1:] = ray_indices[1:] != ray_indices[:-1]
    pos = torch.where(is_start, torch.arange(n), torch.zeros(n, dtype=torch.long))
    return torch.cummax(pos, 0).values


def _excl_seg_sum(v, ray_indices):
    """exclusive prefix sum of v (fp64) inside each contiguous ray segment."""
    c = torch.cumsum(v, 0)
    e = c - v
    return e - e[_seg_start_index(ray_indices)]


def _excl_seg_sum_reverse(v, ray_indices):
    return _excl_seg_sum(v.flip(0), ray_indices.flip(0)).flip(0)


def _excl_seg_prod(v, ray_indices):
    """sequential in fp64 (products of (1-alpha) may hit exact zeros)."""
    out = np.empty(v.shape[0], dtype=np.float64)
    vv, ri = v.numpy(), ray_indices.numpy()
    T = 1.0
    for i in range(vv.shape[0]):
        if i == 0 or ri[i] != ri[i - 1]:
            T = 1.0
        out[i] = T
        T *= vv[i]
    return torch.from_numpy(out)


class _TransFromSigma(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sigmas_dt, ray_indices):
        T = torch.exp(-_excl_seg_sum(sigmas_dt.double().view(-1), ray_indices)).float().view_as(sigmas_dt)
        ctx.save_for_backward(T, ray_indices)
        return T

    @staticmethod
    def backward(ctx, gT):
        T, ray_indices = ctx.saved_tensors
        g = -_excl_seg_sum_reverse((gT * T).double().view(-1), ray_indices)
        return g.float().view_as(T), None


class _TransFromAlpha(torch.autograd.Function):
    @staticmethod
    def forward(ctx, alphas, ray_indices):
        T = _excl_seg_prod(1.0 - alphas.detach().double().view(-1), ray_indices).float().view_as(alphas)
        ctx.save_for_backward(T, alphas, ray_indices)
        return T

    @staticmethod
    def backward(ctx, gT):
        T, alphas, ray_indices = ctx.saved_tensors
        g = -_excl_seg_sum_reverse((gT * T).double().view(-1), ray_indices).view_as(T)
        g = g / (1.0 - alphas.double()).clamp_min(1e-10)
        return g.float(), None


def _check_sorted(ray_indices):
    ri = ray_indices.long().view(-1).cpu()
    assert ri.numel() == 0 or bool((ri[1:] >= ri[:-1]).all()), "ray_indices must be sorted"
    return ri


def render_transmittance_from_density(t_starts, t_ends, sigmas, *, packed_info=None, ray_indices=None, n_rays=None):
    if ray_indices is None:
        ray_indices = unpack_info(packed_info)
    return _TransFromSigma.apply(sigmas * (t_ends - t_starts), _check_sorted(ray_indices))


def render_transmittance_from_alpha(alphas, *, packed_info=None, ray_indices=None, n_rays=None):
    if ray_indices is None:
        ray_indices = unpack_info(packed_info)
    return _TransFromAlpha.apply(alphas, _check_sorted(ray_indices))


def render_weight_from_density(t_starts, t_ends, sigmas, *, packed_info=None, ray_indices=None, n_rays=None):
    T = render_transmittance_from_density(t_starts, t_ends, sigmas, packed_info=packed_info,
                                          ray_indices=ray_indices, n_rays=n_rays)
    alphas = 1.0 - torch.exp(-sigmas * (t_ends - t_starts))
    return T * alphas


def render_weight_from_alpha(alphas, *, packed_info=None, ray_indices=None, n_rays=None):
    T = render_transmittance_from_alpha(alphas, packed_info=packed_info, ray_indices=ray_indices, n_rays=n_rays)
    return T * alphas


@torch.no_grad()
def render_visibility(alphas, *, ray_indices=None, packed_info=None, n_rays=None, early_stop_eps=1e-4,
                      alpha_thre=0.0):
    T = render_transmittance_from_alpha(alphas, packed_info=packed_info, ray_indices=ray_indices, n_rays=n_rays)
    vis = T >= early_stop_eps
    if alpha_thre > 0:
        vis = vis & (alphas >= alpha_thre)
    return vis.view(-1)


def accumulate_along_rays(weights, ray_indices, values=None, n_rays=None):
    assert ray_indices.dim() == 1 and weights.dim() == 2
    src = weights if values is None else weights * values
    if ray_indices.numel() == 0:
        assert n_rays is not None
        return torch.zeros((n_rays, src.shape[-1]), dtype=src.dtype)
    if n_rays is None:
        n_rays = int(ray_indices.max()) + 1
    out = torch.zeros((n_rays, src.shape[-1]), dtype=src.dtype)
    return out.index_add(0, ray_indices.long(), src)


# ------------------------------------------------------------------------------------------------
# ray marching
# ------------------------------------------------------------------------------------------------
@torch.no_grad()
def march_rays_packed(rays_o, rays_d, t_min, t_max, roi, binary, ctype, step, cone_angle):
    """two-pass marcher; returns (packed_info[n,2] int32, ray_indices int64, t_starts[n,1], t_ends[n,1])."""
    o, d, tmin, tmax, r = _f32c(rays_o), _f32c(rays_d), _f32c(t_min), _f32c(t_max), _f32c(roi)
    n = o.shape[0]
    res = torch.tensor(list(binary.shape), dtype=torch.int32)
    g = binary.contiguous().cpu().view(torch.uint8)
    num_steps = torch.zeros(n, dtype=torch.int32)
    args = (ctypes.c_int64(n), _p(o, _f), _p(d, _f), _p(tmin, _f), _p(tmax, _f), _p(r, _f), _p(res, _i32),
            _p(g, _u8), ctypes.c_int(ctype.value), ctypes.c_float(step), ctypes.c_float(cone_angle))
    _lib.nsro_ray_march(*args, None, _p(num_steps, _i32), None, None, None)
    cum = torch.cumsum(num_steps, 0, dtype=torch.int32)
    packed = torch.stack([cum - num_steps, num_steps], dim=-1).contiguous()
    total = int(cum[-1]) if n else 0
    ray_indices = torch.empty(total, dtype=torch.int64)
    t_starts, t_ends = torch.empty(total, 1), torch.empty(total, 1)
    _lib.nsro_ray_march(*args, _p(packed, _i32), None, _p(ray_indices, _i64), _p(t_starts, _f), _p(t_ends, _f))
    return packed, ray_indices, t_starts, t_ends


@torch.no_grad()
def ray_marching(rays_o, rays_d, t_min=None, t_max=None, scene_aabb=None, grid=None, sigma_fn=None,
                 alpha_fn=None, early_stop_eps=1e-4, alpha_thre=0.0, near_plane=None, far_plane=None,
                 render_step_size=1e-3, stratified=False, cone_angle=0.0):
    if sigma_fn is not None and alpha_fn is not None:
        raise ValueError("Only one of `sigma_fn` and `alpha_fn` should be provided.")
    if t_min is None or t_max is None:
        if scene_aabb is not None:
            t_min, t_max = ray_aabb_intersect(rays_o, rays_d, scene_aabb)
        else:
            t_min = torch.zeros_like(rays_o[..., 0])
            t_max = torch.ones_like(rays_o[..., 0]) * 1e10
    if near_plane is not None:
        t_min = torch.clamp(t_min, min=near_plane)
    if far_plane is not None:
        t_max = torch.clamp(t_max, max=far_plane)
    if stratified:
        t_min = t_min + torch.rand_like(t_min) * render_step_size
    if grid is not None:
        roi, binary, ctype = grid.roi_aabb, grid.binary, grid.contraction_type
    else:
        roi = torch.tensor([-1e10, -1e10, -1e10, 1e10, 1e10, 1e10])
        binary = torch.ones([1, 1, 1], dtype=torch.bool)
        ctype = ContractionType.AABB
    _, ray_indices, t_starts, t_ends = march_rays_packed(rays_o, rays_d, t_min, t_max, roi, binary, ctype,
                                                         render_step_size, cone_angle)
    if (alpha_thre > 0.0 or early_stop_eps > 0.0) and (sigma_fn is not None or alpha_fn is not None):
        if grid is not None:
            alpha_thre = min(alpha_thre, grid.occs.mean().item())
        if sigma_fn is not None:
            sigmas = sigma_fn(t_starts, t_ends, ray_indices)
            assert sigmas.shape == t_starts.shape
            alphas = 1.0 - torch.exp(-sigmas * (t_ends - t_starts))
        else:
            alphas = alpha_fn(t_starts, t_ends, ray_indices)
            assert alphas.shape == t_starts.shape
        masks = render_visibility(alphas, ray_indices=ray_indices, early_stop_eps=early_stop_eps,
                                  alpha_thre=alpha_thre, n_rays=rays_o.shape[0])
        ray_indices, t_starts, t_ends = ray_indices[masks], t_starts[masks], t_ends[masks]
    return ray_indices, t_starts, t_ends


from . import intersection  # noqa: E402,F401
