"""CPU restatement of ``torch_efficient_distloss.flatten_eff_distloss`` (sunset1995/torch_efficient_distloss, the
dependency the reference imports at systems/nerf.py:4 and systems/neus.py:4; absent from /root/reference and from this
image: PARITY UNPINNED against the package itself).  TEST INFRASTRUCTURE ONLY.

Two forms: the *definition* of the Mip-NeRF 360 distortion loss (O(n^2) per ray, differentiable through autograd) and
the package's O(n) prefix-sum formulation with its hand-written gradient -- the tests pin one against the other.
"""
import torch


def distortion_definition(w, m, interval, ray_id):
    """1/(max(ray_id)+1) * sum_rays [ sum_ij w_i w_j |m_i - m_j| + 1/3 sum_i w_i^2 interval_i ]   (float64, autograd)"""
    n_rays = int(ray_id.max()) + 1 if ray_id.numel() else 1
    total = w.new_zeros((), dtype=torch.float64)
    for r in range(n_rays):
        sel = ray_id == r
        wr, mr, ir = w[sel].double(), m[sel].double(), interval[sel].double()
        total = total + (wr[:, None] * wr[None, :] * (mr[:, None] - mr[None, :]).abs()).sum() + (wr * wr * ir).sum() / 3.0
    return total / n_rays


def flatten_eff_distloss(w, m, interval, ray_id):
    """the package's formulation: exclusive per-ray prefix sums.  -> (loss, dloss/dw) in float64"""
    w, m, interval = w.double(), m.double(), interval.double()
    n_rays = int(ray_id.max()) + 1
    loss = w.new_zeros(())
    grad = torch.zeros_like(w)
    for r in range(n_rays):
        idx = torch.nonzero(ray_id == r)[:, 0]
        wr, mr, ir = w[idx], m[idx], interval[idx]
        wm = wr * mr
        w_pre, wm_pre = torch.cumsum(wr, 0) - wr, torch.cumsum(wm, 0) - wm
        w_suf, wm_suf = wr.sum() - w_pre - wr, wm.sum() - wm_pre - wm
        loss = loss + (2 * wr * (mr * w_pre - wm_pre)).sum() + (ir * wr * wr).sum() / 3.0
        grad[idx] = 2 * (mr * (w_pre - w_suf) + (wm_suf - wm_pre)) + 2.0 / 3.0 * ir * wr
    return loss / n_rays, grad / n_rays
