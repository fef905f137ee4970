"""``nerfacc.ray_marching`` (reference call sites models/nerf.py:83, models/neus.py:159,210)."""
import torch

from nsr_hip import ops as _ops

from .contraction import ContractionType
from .intersection import ray_aabb_intersect
from .vol_rendering import render_visibility


@torch.no_grad()
def ray_marching(rays_o, rays_d, t_min=None, t_max=None, scene_aabb=None, grid=None, sigma_fn=None, alpha_fn=None,
                 early_stop_eps=1e-4, alpha_thre=0.0, near_plane=None, far_plane=None, render_step_size=1e-3,
                 stratified=False, cone_angle=0.0):
    """-> (ray_indices int64 [n], t_starts [n,1], t_ends [n,1]), ray-major, ascending t inside a ray.

    Two host syncs remain by construction of the API (data-dependent output shapes): the marched sample
    count, and -- only when ``sigma_fn``/``alpha_fn`` prune invisible samples -- the kept count.
    """
    if not rays_o.is_cuda:
        raise NotImplementedError("nerfacc(gfx950): only GPU tensors are supported")
    if alpha_fn is not None and sigma_fn is not None:
        raise ValueError("Only one of `alpha_fn` and `sigma_fn` should be provided.")
    rays_o, rays_d = rays_o.float().contiguous(), rays_d.float().contiguous()
    n_rays = rays_o.shape[0]
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
    roi_host = None
    if grid is not None:
        roi, binary, ctype = grid.roi_aabb, grid.binary, grid.contraction_type
        roi_host = getattr(grid, "_roi_host", None)
    else:
        roi = torch.tensor([-1e10, -1e10, -1e10, 1e10, 1e10, 1e10], dtype=torch.float32, device=rays_o.device)
        binary = torch.ones([1, 1, 1], dtype=torch.bool, device=rays_o.device)
        ctype = ContractionType.AABB
    packed, ray_indices, t_starts, t_ends = _ops.ray_march(
        rays_o, rays_d, t_min.float().contiguous(), t_max.float().contiguous(), roi.float().contiguous(),
        binary.contiguous(), ctype.value, render_step_size, cone_angle, roi_host=roi_host)
    ray_indices._nsr_packed = (n_rays, ray_indices.shape[0], packed)

    if (alpha_thre > 0.0 or early_stop_eps > 0.0) and (sigma_fn is not None or alpha_fn is not None):
        if grid is not None and alpha_thre > 0.0:  # min(alpha_thre<=0, mean>=0) disables the test: skip the sync
            alpha_thre = min(alpha_thre, grid.occs.mean().item())
        if sigma_fn is not None:
            sigmas = sigma_fn(t_starts, t_ends, ray_indices)
            assert sigmas.shape == t_starts.shape, f"sigmas must have shape of (N, 1)! Got {sigmas.shape}"
            alphas = 1.0 - torch.exp(-sigmas * (t_ends - t_starts))
        else:
            alphas = alpha_fn(t_starts, t_ends, ray_indices)
            assert alphas.shape == t_starts.shape, f"alphas must have shape of (N, 1)! Got {alphas.shape}"
        masks = render_visibility(alphas, packed_info=packed, early_stop_eps=early_stop_eps, alpha_thre=alpha_thre)
        ray_indices, t_starts, t_ends = _ops.compact_samples(masks.contiguous(), ray_indices, t_starts, t_ends)
    return ray_indices, t_starts, t_ends
