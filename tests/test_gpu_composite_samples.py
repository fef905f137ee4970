"""The NeRF step's compositing pair (csrc/fused.hip k_composite_forward_samples / k_composite_backward_samples: one lane per
kept sample, a wave per 64 samples) DIRECTLY against the oracle: nerfacc's render_weight_from_density + accumulate_along_rays
(oracle/nerfacc_ref, fp64 segmented scans with nerfacc's backward formulas; reference models/nerf.py:105-108), trunc_exp and
the density bias of models/geometry.py:122-156, and the system's masked smooth-L1 (systems/nerf.py:97) through CPU autograd.
Tolerances: weights / transmittance rtol 1e-4 (fp32 scans against fp64), per-ray outputs 5e-6 absolute, gradients 2e-4 of
their maximum."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(n_rays, max_count, seed, long_every=0, empty_every=7):
    g = torch.Generator().manual_seed(seed)
    counts = torch.randint(0, max_count, (n_rays,), generator=g)
    if empty_every:
        counts[::empty_every] = 0
    if long_every:
        counts[3::long_every] = torch.randint(65, 400, counts[3::long_every].shape, generator=g)  # rays spanning chunks
    starts = torch.cumsum(counts, 0) - counts
    n = int(counts.sum())
    m = max(n, 1)
    return dict(packed=torch.stack([starts, counts], 1).int(), n=n, m=m, g=g,
                ray_idx=torch.repeat_interleave(torch.arange(n_rays), counts),
                logit=(torch.randn(m, generator=g) * 2 - 1).half(), rgb=torch.rand(m, 3, generator=g).half(),
                t0=torch.rand(m, generator=g), bg=torch.tensor([1.0, 0.5, 0.25]), gt=torch.rand(n_rays, 3, generator=g))


def _oracle(c, n_rays, bias, mode, up, scale):
    from oracle import nerfacc_ref as R
    n = c["n"]
    logit = c["logit"][:n].float().clone().requires_grad_(True)
    rgb = c["rgb"][:n].float().clone().requires_grad_(True)
    t0, t1 = c["t0"][:n].view(-1, 1), c["t0"][:n].view(-1, 1) + 0.01
    sigma = torch.exp(logit + bias).view(-1, 1)
    ri = c["ray_idx"]
    w = R.render_weight_from_density(t0, t1, sigma, ray_indices=ri, n_rays=n_rays)
    T = R.render_transmittance_from_density(t0, t1, sigma, ray_indices=ri, n_rays=n_rays)
    op = R.accumulate_along_rays(w, ri, None, n_rays)
    dp = R.accumulate_along_rays(w, ri, (t0 + t1) / 2, n_rays)
    comp = R.accumulate_along_rays(w, ri, rgb, n_rays) + c["bg"] * (1.0 - op)
    valid = op[:, 0] > 0
    if mode == "upstream":
        loss = (comp * up["c"]).sum() + (op[:, 0] * up["o"]).sum() + (dp[:, 0] * up["d"]).sum() + (w[:, 0] * up["w"][:n]).sum()
        acc = None
    else:
        per = torch.nn.functional.smooth_l1_loss(comp[valid], c["gt"][valid], reduction="sum")
        nv = int(valid.sum())
        loss = scale * per / max(3 * nv, 1)
        acc = (float(per), nv)
    if n:
        loss.backward()
    return dict(w=w.detach().view(-1), tr=T.detach().view(-1), rgb=comp.detach(), op=op.detach().view(-1),
                dp=dp.detach().view(-1), d_rgb=rgb.grad if n else torch.zeros(0, 3),
                d_logit=logit.grad if n else torch.zeros(0), acc=acc)


@pytest.mark.parametrize("n_rays,max_count,long_every", [(8192, 40, 0), (8192, 30, 9), (1147, 40, 5), (3, 40, 0), (9, 40, 2),
                                                         (64, 40, 1), (5, 1, 0)])
@pytest.mark.parametrize("mode", ["folded", "acc", "upstream"])
def test_sample_partitioned_compositing_matches_the_oracle(n_rays, max_count, long_every, mode):
    from nsr_hip import check, lib, ptr, stream_ptr
    c = _case(n_rays, max_count, 100 + n_rays + long_every, long_every)
    n, m, g = c["n"], c["m"], c["g"]
    bias, scale = -1.0, 2.0
    up = dict(c=torch.randn(n_rays, 3, generator=g) * 0.1, o=torch.randn(n_rays, generator=g) * 0.1,
              d=torch.randn(n_rays, generator=g) * 0.1, w=torch.randn(m, generator=g) * 0.1)
    want = _oracle(c, n_rays, bias, mode, up, scale)
    cap = m + 77  # capacity of the sample arrays > live samples: the live count comes from the device
    out1 = torch.zeros(cap, 16).half(); out1[:m, 0] = c["logit"]
    out2 = torch.zeros(cap, 16).half(); out2[:m, :3] = c["rgb"]
    t0 = torch.zeros(cap); t0[:m] = c["t0"]
    ri = torch.full((cap,), 2 ** 40, dtype=torch.int64); ri[:n] = c["ray_idx"]  # rows behind the live count: never read
    out1, out2, t0, ri = out1.cuda(), out2.cuda(), t0.cuda(), ri.cuda()
    t1 = t0 + 0.01
    packed, bg, gt = c["packed"].cuda(), c["bg"].cuda(), c["gt"].cuda()
    n_dev = torch.tensor([n], dtype=torch.int32).cuda()
    upc = {k: v.cuda() for k, v in up.items()}
    upw = torch.zeros(cap).cuda(); upw[:m] = upc["w"]
    w, tr = torch.full((cap,), -7.0).cuda(), torch.full((cap,), -7.0).cuda()
    rgb, op, dp = torch.full((n_rays, 3), -7.0).cuda(), torch.full((n_rays,), -7.0).cuda(), torch.full((n_rays,), -7.0).cuda()
    acc = torch.full((2,), -1.0).cuda()
    d_rgb, d_logit = torch.full((cap, 3), -7.0).cuda(), torch.full((cap,), -7.0).cuda()
    part = torch.full((int(lib.nsr_composite_l1_partials_floats(n_rays)),), float("nan")).cuda()
    s = stream_ptr()
    folded = mode == "folded"
    check(lib.nsr_composite_forward_samples(ptr(out1), 16, bias, ptr(t0), ptr(t1), ptr(out2), 16, ptr(packed), ptr(ri), ptr(bg),
                                            ptr(w), ptr(tr), ptr(rgb), ptr(op), ptr(dp), ptr(gt) if folded else None,
                                            ptr(part) if folded else None, n_rays, cap, ptr(n_dev), s), "fwd")
    if mode == "acc":
        check(lib.nsr_smooth_l1_valid_set(ptr(rgb), ptr(op), ptr(gt), ptr(acc), n_rays, s), "l1")
    if mode == "upstream":
        check(lib.nsr_composite_backward_samples(ptr(out1), 16, bias, ptr(t0), ptr(t1), ptr(out2), 16, ptr(packed), ptr(ri),
                                                 ptr(bg), ptr(w), ptr(tr), ptr(upc["c"]), ptr(upc["o"]), ptr(upc["d"]), ptr(upw),
                                                 None, None, None, None, None, 1.0, ptr(d_rgb), ptr(d_logit), n_rays, cap,
                                                 ptr(n_dev), s), "bwd")
    else:
        check(lib.nsr_composite_backward_samples(ptr(out1), 16, bias, ptr(t0), ptr(t1), ptr(out2), 16, ptr(packed), ptr(ri),
                                                 ptr(bg), ptr(w), ptr(tr), None, None, None, None, ptr(rgb), ptr(op), ptr(gt),
                                                 ptr(part) if folded else None, ptr(acc), scale, ptr(d_rgb), ptr(d_logit),
                                                 n_rays, cap, ptr(n_dev), s), "bwd")
    torch.cuda.synchronize()
    assert torch.allclose(w[:n].cpu(), want["w"], rtol=1e-4, atol=1e-7), float((w[:n].cpu() - want["w"]).abs().max())
    assert torch.allclose(tr[:n].cpu(), want["tr"], rtol=1e-4, atol=1e-7)
    for k, got in (("rgb", rgb), ("op", op), ("dp", dp)):
        assert torch.allclose(got.cpu(), want[k], rtol=1e-5, atol=5e-6), (k, float((got.cpu() - want[k]).abs().max()))
    # rows behind the live count are never written
    assert float(w[n:].min()) == -7.0 and float(d_logit[n:].min()) == -7.0 and float(d_rgb[n:].min()) == -7.0
    # rays without samples: background, opacity 0 -- written, not left over
    empty = (c["packed"][:, 1] == 0).cuda()
    assert bool((op[empty] == 0).all()) and bool((rgb[empty] == bg).all()) and bool((dp[empty] == 0).all())
    if n:
        for k, got in (("d_rgb", d_rgb[:n]), ("d_logit", d_logit[:n])):
            sc = float(want[k].abs().max()) + 1e-12
            err = float((got.cpu() - want[k]).abs().max())
            assert err <= 2e-4 * sc, (k, err, sc)
        assert float(want["d_logit"].abs().max()) > 0
    if mode != "upstream":
        assert float(acc[1]) == want["acc"][1]
        assert abs(float(acc[0]) - want["acc"][0]) <= 1e-5 * abs(want["acc"][0]) + 1e-6


def test_overflowed_density_does_not_poison_the_ray():
    """exp(logit) = inf: T = 0 behind the sample (nerfacc's sequential loop), never NaN -- also across a chunk boundary"""
    from nsr_hip import check, lib, ptr, stream_ptr
    packed = torch.tensor([[0, 150]], dtype=torch.int32).cuda()
    ri = torch.zeros(150, dtype=torch.int64).cuda()
    out1 = torch.zeros(150, 16).half().cuda()
    out1[40, 0] = 200.0
    out2 = torch.rand(150, 16).half().cuda()
    t0 = torch.arange(150).float().cuda() * 0.01
    t1 = t0 + 0.01
    bg = torch.zeros(3).cuda()
    w, tr = torch.zeros(150).cuda(), torch.zeros(150).cuda()
    rgb, op, dp = torch.zeros(1, 3).cuda(), torch.zeros(1).cuda(), torch.zeros(1).cuda()
    check(lib.nsr_composite_forward_samples(ptr(out1), 16, 0.0, ptr(t0), ptr(t1), ptr(out2), 16, ptr(packed), ptr(ri), ptr(bg),
                                            ptr(w), ptr(tr), ptr(rgb), ptr(op), ptr(dp), None, None, 1, 150, None, stream_ptr()),
          "fwd")
    assert bool(torch.isfinite(w).all()) and bool(torch.isfinite(rgb).all()) and float(tr[41:].abs().max()) == 0.0
