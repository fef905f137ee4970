"""``nerfacc`` volumetric rendering ops (reference call sites models/nerf.py:105-108, models/neus.py:181-184,
237-242): per-ray transmittance scans and per-ray accumulation on libnsr_hip.so (one wavefront per ray)."""
import torch

from nsr_hip import ops as _ops

from .pack import cached_packed_info, pack_info


def _packed(packed_info, ray_indices, n_rays):
    if packed_info is not None:
        return packed_info.int().contiguous(), packed_info.shape[0]
    if ray_indices is None:
        raise ValueError("Either `packed_info` or `ray_indices` must be given")
    if n_rays is None:
        n_rays = int(ray_indices.max()) + 1 if ray_indices.numel() else 0
    return cached_packed_info(ray_indices, int(n_rays)), int(n_rays)


def render_transmittance_from_density(t_starts, t_ends, sigmas, *, packed_info=None, ray_indices=None, n_rays=None):
    packed, n_rays = _packed(packed_info, ray_indices, n_rays)
    return _ops.transmittance_from_sigma(sigmas, t_starts.float(), t_ends.float(), packed, n_rays)


def render_transmittance_from_alpha(alphas, *, packed_info=None, ray_indices=None, n_rays=None):
    packed, n_rays = _packed(packed_info, ray_indices, n_rays)
    return _ops.transmittance_from_alpha(alphas, packed, n_rays)


def render_weight_from_density(t_starts, t_ends, sigmas, *, packed_info=None, ray_indices=None, n_rays=None):
    """w_i = T_i * (1 - exp(-sigma_i * (t1_i - t0_i))),  T_i = exp(-sum_{j<i} sigma_j dt_j)  -> [n,1]"""
    trans = render_transmittance_from_density(t_starts, t_ends, sigmas, packed_info=packed_info,
                                              ray_indices=ray_indices, n_rays=n_rays)
    alphas = 1.0 - torch.exp(-sigmas * (t_ends - t_starts))
    return trans * alphas


def render_weight_from_alpha(alphas, *, packed_info=None, ray_indices=None, n_rays=None):
    """w_i = T_i * alpha_i,  T_i = prod_{j<i} (1 - alpha_j)  -> [n,1]"""
    trans = render_transmittance_from_alpha(alphas, packed_info=packed_info, ray_indices=ray_indices, n_rays=n_rays)
    return trans * alphas


@torch.no_grad()
def render_visibility(alphas, *, ray_indices=None, packed_info=None, n_rays=None, early_stop_eps=1e-4,
                      alpha_thre=0.0):
    trans = render_transmittance_from_alpha(alphas, packed_info=packed_info, ray_indices=ray_indices, n_rays=n_rays)
    vis = trans >= early_stop_eps
    if alpha_thre > 0:
        vis = vis & (alphas >= alpha_thre)
    return vis.squeeze(-1)


def accumulate_along_rays(weights, ray_indices, values=None, n_rays=None):
    """out[n_rays, D] = sum over each ray's samples of w * v (D=1, v=1 when ``values`` is None)."""
    if ray_indices.dim() != 1 or weights.dim() != 2:
        raise ValueError("accumulate_along_rays: ray_indices must be [n] and weights [n,1]")
    if values is not None and (values.dim() != 2 or values.shape[0] != weights.shape[0]):
        raise ValueError("accumulate_along_rays: values must be [n, D]")
    if n_rays is None:
        if ray_indices.numel() == 0:
            raise ValueError("accumulate_along_rays: n_rays is required when there are no samples")
        n_rays = int(ray_indices.max()) + 1
    dim = 1 if values is None else values.shape[-1]
    if ray_indices.numel() == 0:
        return torch.zeros((n_rays, dim), device=weights.device, dtype=weights.dtype)
    packed = cached_packed_info(ray_indices, int(n_rays))
    return _ops.accumulate_along_rays(weights, values, ray_indices.long().contiguous(), packed, int(n_rays))
