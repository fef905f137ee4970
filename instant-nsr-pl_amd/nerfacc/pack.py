"""``pack_info`` / ``unpack_info``: (start, count) per ray <-> sorted per-sample ray indices."""
import torch

from nsr_hip import ops as _ops


@torch.no_grad()
def pack_info(ray_indices, n_rays=None):
    if n_rays is None:
        n_rays = int(ray_indices.max()) + 1 if ray_indices.numel() else 0
    return _ops.pack_info(ray_indices.long().contiguous(), int(n_rays))


@torch.no_grad()
def unpack_info(packed_info, n_samples=None):
    cnt = packed_info[:, 1].long()
    return torch.repeat_interleave(torch.arange(packed_info.shape[0], device=packed_info.device), cnt)


def cached_packed_info(ray_indices, n_rays):
    """``ray_marching`` tags the ray_indices it returns with their packed_info so the 4-5 compositing calls
    of one forward (reference models/nerf.py:105-108) do not rebuild it."""
    tag = getattr(ray_indices, "_nsr_packed", None)
    if tag is not None and tag[0] == n_rays and tag[1] == ray_indices.shape[0]:
        return tag[2]
    packed = pack_info(ray_indices, n_rays)
    try:
        ray_indices._nsr_packed = (n_rays, ray_indices.shape[0], packed)
    except Exception:  # noqa: BLE001
        pass
    return packed
