"""HIP distortion loss (drop-in ``torch_efficient_distloss.flatten_eff_distloss``) against the CPU oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(counts, seed):
    g = torch.Generator().manual_seed(seed)
    ray_id = torch.cat([torch.full((c,), r, dtype=torch.int64) for r, c in enumerate(counts)])
    n = ray_id.numel()
    w = torch.rand(n, generator=g) * 0.3
    m = torch.cat([torch.sort(torch.rand(c, generator=g) * 4 + 0.5).values for c in counts if c])
    interval = torch.rand(n, generator=g) * 0.05 + 0.01
    return w, m, interval, ray_id


@pytest.mark.parametrize("counts", [[5, 1, 17, 64, 3], [0, 130, 0, 65, 1, 0, 200], [1], [1027, 2, 1027]])
def test_flatten_eff_distloss_matches_oracle(counts):
    from oracle import distloss_ref
    from torch_efficient_distloss import flatten_eff_distloss
    w, m, interval, ray_id = _case(counts, 0)
    ref_loss, ref_grad = distloss_ref.flatten_eff_distloss(w, m, interval, ray_id)
    wg = w.cuda().requires_grad_(True)
    loss = flatten_eff_distloss(wg, m.cuda(), interval.cuda(), ray_id.cuda())
    (loss * 3.0).backward()
    assert abs(float(loss) - float(ref_loss)) <= 2e-5 * max(1.0, abs(float(ref_loss)))
    assert torch.allclose(wg.grad.cpu().double(), 3.0 * ref_grad, rtol=2e-4, atol=1e-6)


def test_flatten_eff_distloss_on_marched_samples():
    """on real ray-packed samples of the drop-in nerfacc (the call of systems/nerf.py:104), incl. an [n,1] weights view"""
    import nerfacc
    from oracle import distloss_ref
    from torch_efficient_distloss import flatten_eff_distloss
    g = torch.Generator().manual_seed(1)
    o = torch.nn.functional.normalize(torch.randn(64, 3, generator=g), dim=-1) * 3.0
    d = torch.nn.functional.normalize(-o + torch.randn(64, 3, generator=g) * 0.2, dim=-1)
    aabb = torch.tensor([-1.0, -1, -1, 1, 1, 1])
    ri, t0, t1 = nerfacc.ray_marching(o.cuda(), d.cuda(), scene_aabb=aabb.cuda(), render_step_size=0.02)
    n = ri.numel()
    assert n > 500
    w = (torch.rand(n, 1, generator=g) * 0.1).cuda().requires_grad_(True)
    mid, dt = ((t0 + t1) / 2).view(-1), (t1 - t0).view(-1)
    loss = flatten_eff_distloss(w.view(-1), mid, dt, ri)
    loss.backward()
    ref_loss, ref_grad = distloss_ref.flatten_eff_distloss(w.detach().cpu().view(-1), mid.cpu(), dt.cpu(), ri.cpu())
    assert abs(float(loss) - float(ref_loss)) <= 2e-5 * max(1.0, abs(float(ref_loss)))
    assert torch.allclose(w.grad.cpu().view(-1).double(), ref_grad, rtol=2e-4, atol=1e-7)
    assert float(flatten_eff_distloss(w[:0].view(-1), mid[:0], dt[:0], ri[:0])) == 0.0
