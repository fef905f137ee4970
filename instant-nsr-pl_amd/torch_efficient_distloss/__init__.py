"""Drop-in for the ``torch_efficient_distloss`` import of the reference (``systems/nerf.py:4,103-106``,
``systems/neus.py:4,131-139``): ``flatten_eff_distloss(w, m, interval, ray_id)``, the Mip-NeRF 360 distortion loss over
ray-packed samples, on the HIP segmented-scan kernels of ``csrc/render.hip`` (one wavefront per ray, no atomics).

    loss = 1 / (max(ray_id) + 1) * sum_rays [ sum_ij w_i w_j |m_i - m_j| + 1/3 sum_i w_i^2 interval_i ]

Like the package it replaces, the gradient flows to ``w`` only, and the mean is over ``ray_id.max() + 1`` rays.
"""
import torch

from nerfacc.pack import cached_packed_info
from nsr_hip import check, lib, ptr, stream_ptr

__all__ = ["flatten_eff_distloss"]


class _FlattenEffDistLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, m, interval, ray_id):
        if not w.is_cuda:
            raise RuntimeError("flatten_eff_distloss (gfx950): GPU tensors only -- there is no CPU path in the product")
        n = w.numel()
        if n == 0:
            ctx.empty = True
            return w.new_zeros(())
        ctx.empty = False
        n_rays = int(ray_id.max().item()) + 1  # the divisor torch_efficient_distloss uses (trailing empty rays are not counted)
        packed = cached_packed_info(ray_id.reshape(-1).contiguous(), n_rays)
        w32, m32 = w.detach().reshape(-1).float().contiguous(), m.detach().reshape(-1).float().contiguous()
        i32 = interval.detach().reshape(-1).float().contiguous()
        if i32.numel() == 1 and n > 1:
            i32 = i32.expand(n).contiguous()
        ray_loss = torch.empty(n_rays, dtype=torch.float32, device=w.device)
        with torch.cuda.device(w.device):
            check(lib.nsr_distortion_loss_forward(ptr(packed), ptr(w32), ptr(m32), ptr(i32), ptr(ray_loss), n_rays,
                                                  stream_ptr()), "nsr_distortion_loss_forward")
        ctx.save_for_backward(packed, w32, m32, i32)
        ctx.n_rays, ctx.shape, ctx.dtype = n_rays, w.shape, w.dtype
        return (ray_loss.sum() / n_rays).to(w.dtype)

    @staticmethod
    def backward(ctx, grad_out):
        if ctx.empty:
            return None, None, None, None
        packed, w32, m32, i32 = ctx.saved_tensors
        g = torch.empty_like(w32)
        with torch.cuda.device(w32.device):
            check(lib.nsr_distortion_loss_backward(ptr(packed), ptr(w32), ptr(m32), ptr(i32), ptr(g), ctx.n_rays,
                                                   stream_ptr()), "nsr_distortion_loss_backward")
        return (g * (grad_out.float() / ctx.n_rays)).view(ctx.shape).to(ctx.dtype), None, None, None


def flatten_eff_distloss(w, m, interval, ray_id):
    """w, m, interval: [n] weights, midpoints and interval lengths of ray-packed samples (sorted along each ray);
    ray_id: [n] int64 ray of every sample (non-decreasing).  Returns the scalar loss."""
    return _FlattenEffDistLoss.apply(w, m, interval, ray_id)
