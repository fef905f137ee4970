"""``nerfacc.intersection.ray_aabb_intersect`` (reference call site models/neus.py:12,153)."""
import torch

from nsr_hip import ops as _ops


@torch.no_grad()
def ray_aabb_intersect(rays_o, rays_d, aabb):
    """-> (t_min[n_rays], t_max[n_rays]); a miss gives 1e10 for both (the reference tests ``t_max > 1e9``,
    models/neus.py:155-157); t_min is clamped to >= 0."""
    if rays_o.dim() != 2 or rays_o.shape[-1] != 3 or rays_o.shape != rays_d.shape or aabb.numel() != 6:
        raise ValueError("ray_aabb_intersect: rays_o/rays_d must be [n_rays,3] and aabb [6]")
    return _ops.ray_aabb_intersect(rays_o.float().contiguous(), rays_d.float().contiguous(),
                                   aabb.float().contiguous())
