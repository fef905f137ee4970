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


def test_overflowed_density_does_not_poison_the_ray():
    """sigma = inf (trunc_exp of a large logit overflows fp32) on consecutive samples: nerfacc's sequential loop gives
    alpha = 1, T = 0 behind the sample and finite weights; a scan written as `inclusive - own` would produce inf - inf = NaN
    for every later sample of the ray (seen once as a NaN pixel in an eval render)"""
    import nerfacc as A
    n_rays, per = 3, 70
    ri = torch.repeat_interleave(torch.arange(n_rays), per).cuda()
    t0 = (torch.arange(per).float() * 0.01).repeat(n_rays).view(-1, 1).cuda()
    t1 = t0 + 0.01
    sig = torch.full((n_rays * per, 1), 2.0).cuda()
    sig[per + 10] = float("inf")
    sig[per + 11] = float("inf")          # two in a row, inside one wavefront
    sig[2 * per + 63] = float("inf")
    sig[2 * per + 64] = float("inf")      # across the 64-sample chunk boundary
    w = A.render_weight_from_density(t0, t1, sig, ray_indices=ri, n_rays=n_rays)
    assert bool(torch.isfinite(w).all())
    w = w.view(n_rays, per)
    assert float(w[1, 11:].abs().max()) == 0.0 and float(w[2, 64:].abs().max()) == 0.0   # nothing behind an opaque sample
    assert abs(float(w[1].sum()) - 1.0) < 1e-5 and abs(float(w[2].sum()) - 1.0) < 1e-5     # the ray is fully absorbed
    assert torch.allclose(w[0], w[1].new_tensor([(1 - torch.exp(torch.tensor(-0.02))) * torch.exp(torch.tensor(-0.02 * k))
                                                 for k in range(per)]), rtol=1e-4)
    # the fused NeRF compositing kernel (logits; exp(logit) overflows above 88.7)
    from nsr_hip import check, lib, ptr, stream_ptr
    out1 = torch.zeros((n_rays * per, 16), dtype=torch.float16, device="cuda")
    out1[per + 10, 0] = 100.0
    out1[per + 11, 0] = 100.0
    rgb = torch.full((n_rays * per, 16), 0.5, dtype=torch.float16, device="cuda")
    packed = torch.tensor([[k * per, per] for k in range(n_rays)], dtype=torch.int32, device="cuda")
    bg = torch.ones(3, device="cuda")
    wts, tr = torch.empty(n_rays * per, device="cuda"), torch.empty(n_rays * per, device="cuda")
    comp, op, dp = torch.empty((n_rays, 3), device="cuda"), torch.empty(n_rays, device="cuda"), torch.empty(n_rays, device="cuda")
    check(lib.nsr_composite_forward(ptr(out1), 16, 0.0, ptr(t0), ptr(t1), ptr(rgb), 16, ptr(packed), ptr(bg), ptr(wts), ptr(tr),
                                    ptr(comp), ptr(op), ptr(dp), n_rays, stream_ptr()), "nsr_composite_forward")
    assert bool(torch.isfinite(wts).all()) and bool(torch.isfinite(comp).all()) and abs(float(op[1]) - 1.0) < 1e-5
