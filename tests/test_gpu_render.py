"""HIP segmented scans / accumulation / NeuS alpha vs the oracle (fp64 scans), through the C ABI."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _segments(n_rays, seed=0, max_len=1500):
    g = torch.Generator().manual_seed(seed)
    cnt = torch.randint(0, max_len, (n_rays,), generator=g)
    cnt[::7] = 0      # rays without samples
    cnt[3] = 1        # single-sample ray
    cnt[5] = 64       # exactly one wavefront
    cnt[6] = 65
    ri = torch.repeat_interleave(torch.arange(n_rays), cnt)
    n = ri.numel()
    t0 = torch.rand(n, 1, generator=g)
    t1 = t0 + torch.rand(n, 1, generator=g) * 0.01
    return ri, t0, t1, g


def test_weight_from_density_fwd_bwd():
    from oracle import nerfacc_ref as N
    import nerfacc as A
    ri, t0, t1, g = _segments(200)
    sig = torch.rand(ri.numel(), 1, generator=g) * 30
    gw = torch.randn(ri.numel(), 1, generator=g)
    s_ref = sig.clone().requires_grad_(True)
    w_ref = N.render_weight_from_density(t0, t1, s_ref, ray_indices=ri, n_rays=200)
    w_ref.backward(gw)
    s = sig.cuda().requires_grad_(True)
    w = A.render_weight_from_density(t0.cuda(), t1.cuda(), s, ray_indices=ri.cuda(), n_rays=200)
    w.backward(gw.cuda())
    assert torch.allclose(w.detach().cpu(), w_ref.detach(), rtol=1e-4, atol=1e-6)
    assert torch.allclose(s.grad.cpu(), s_ref.grad, rtol=1e-3, atol=1e-6)
    # sum of weights + final transmittance = 1 (property, size independent)
    opac = A.accumulate_along_rays(w.detach(), ri.cuda(), None, 200).cpu()
    assert bool((opac <= 1 + 1e-5).all()) and bool((opac >= 0).all())


def test_weight_from_alpha_fwd_bwd_and_visibility():
    from oracle import nerfacc_ref as N
    import nerfacc as A
    ri, t0, t1, g = _segments(150, seed=3)
    al = torch.rand(ri.numel(), 1, generator=g) * 0.2
    al[10] = 1.0  # fully opaque sample: T becomes exactly 0 afterwards, backward divides by max(1-a,1e-10)
    gw = torch.randn(ri.numel(), 1, generator=g)
    a_ref = al.clone().requires_grad_(True)
    w_ref = N.render_weight_from_alpha(a_ref, ray_indices=ri, n_rays=150)
    w_ref.backward(gw)
    a = al.cuda().requires_grad_(True)
    w = A.render_weight_from_alpha(a, ray_indices=ri.cuda(), n_rays=150)
    w.backward(gw.cuda())
    assert torch.allclose(w.detach().cpu(), w_ref.detach(), rtol=1e-4, atol=1e-6)
    assert torch.allclose(a.grad.cpu(), a_ref.grad, rtol=2e-3, atol=1e-5)
    vis_ref = N.render_visibility(al, ray_indices=ri, early_stop_eps=1e-2)
    vis = A.render_visibility(al.cuda(), ray_indices=ri.cuda(), n_rays=150, early_stop_eps=1e-2).cpu()
    assert float((vis != vis_ref).float().mean()) < 1e-3  # only T within an ulp of the threshold may flip


@pytest.mark.parametrize("dim", [None, 1, 3, 7])
def test_accumulate_fwd_bwd(dim):
    from oracle import nerfacc_ref as N
    import nerfacc as A
    ri, t0, t1, g = _segments(120, seed=5, max_len=300)
    n = ri.numel()
    w = torch.rand(n, 1, generator=g)
    v = None if dim is None else torch.randn(n, dim, generator=g)
    go = torch.randn(120, 1 if dim is None else dim, generator=g)
    w_ref = w.clone().requires_grad_(True)
    v_ref = None if v is None else v.clone().requires_grad_(True)
    N.accumulate_along_rays(w_ref, ri, v_ref, 120).backward(go)
    wg = w.cuda().requires_grad_(True)
    vg = None if v is None else v.cuda().requires_grad_(True)
    out = A.accumulate_along_rays(wg, ri.cuda(), vg, 120)
    out.backward(go.cuda())
    ref = N.accumulate_along_rays(w, ri, v, 120)
    assert torch.allclose(out.detach().cpu(), ref, rtol=1e-4, atol=1e-4)
    assert torch.allclose(wg.grad.cpu(), w_ref.grad, rtol=1e-4, atol=1e-5)
    if v is not None:
        assert torch.allclose(vg.grad.cpu(), v_ref.grad, rtol=1e-5, atol=1e-6)


def test_neus_alpha_fwd_bwd():
    from oracle import glue_ref
    from nsr_hip import ops
    g = torch.Generator().manual_seed(2)
    n = 5000
    sdf = torch.randn(n, generator=g) * 0.05
    normal = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    dists = torch.rand(n, generator=g) * 0.01
    inv_s = torch.tensor(20.0855)
    for anneal in (0.0, 0.37, 1.0):
        a = [t.clone().requires_grad_(True) for t in (sdf, normal, inv_s)]
        ref = glue_ref.neus_alpha(a[0], a[1], dirs, dists, a[2], anneal)
        ga = torch.randn(n, generator=g)
        ref.backward(ga)
        b = [t.cuda().requires_grad_(True) for t in (sdf, normal, inv_s)]
        out = ops.neus_alpha(b[0], b[1], dirs.cuda(), dists.cuda(), b[2], anneal)
        out.backward(ga.cuda())
        assert torch.allclose(out.detach().cpu(), ref.detach(), rtol=1e-4, atol=1e-6)
        assert torch.allclose(b[0].grad.cpu(), a[0].grad, rtol=2e-3, atol=1e-4)
        assert torch.allclose(b[1].grad.cpu(), a[1].grad, rtol=2e-3, atol=1e-5)
        assert torch.allclose(b[2].grad.cpu(), a[2].grad, rtol=2e-3, atol=1e-4)
