"""(1) the fused AdamW kernel against ``torch.optim.AdamW`` with the reference's optimizer / scheduler settings
(systems/utils.py:314-325, configs/nerf-blender.yaml:74-85: lr 0.01, betas (0.9, 0.99), eps 1e-15, default weight
decay 0.01, MultiStepLR gamma 0.33); (2) the drop-in packages under Lightning's ``precision: 16`` protocol
(configs/nerf-blender.yaml:103): ``torch.autocast`` + ``GradScaler(init_scale=65536)``; (3) non-finite gradients reach
the parameters as inf/NaN so the scaler backs off (tcnn's fp16 atomics overflow the same way)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("device_schedule", [False, True])
def test_adamw_matches_torch(device_schedule):
    from nsr_hip import ops
    torch.manual_seed(0)
    n = 3072 + 40000 + 3  # an MLP slice, a table slice, a ragged tail
    p_ref = torch.nn.Parameter((torch.randn(n, device="cuda") * 0.1))
    p = p_ref.detach().clone()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    shadow = torch.empty(n, dtype=torch.float16, device="cuda")
    grad = torch.zeros_like(p)
    milestones, gamma = (3, 6, 8), 0.33
    opt = torch.optim.AdamW([p_ref], lr=0.01, betas=(0.9, 0.99), eps=1e-15)  # weight_decay default 0.01
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=list(milestones), gamma=gamma)
    step_dev = torch.zeros(1, dtype=torch.int32, device="cuda")
    hyper = torch.zeros(8, dtype=torch.float32, device="cuda")
    for step in range(1, 11):
        g = torch.randn(n, device="cuda") * (10.0 ** torch.randint(-6, 1, (n,), device="cuda").float())
        g[::7] = 0.0  # untouched table entries: eps = 1e-15 makes m / (sqrt(v) + eps) delicate
        p_ref.grad = g.clone()
        opt.step()
        sched.step()
        grad.copy_(g)
        lr = 0.01 * gamma ** sum((step - 1) >= ms for ms in milestones)  # the lr the reference applies at this step
        if device_schedule:
            ops.adam_tick(step_dev, hyper, 0.01, 0.9, 0.99, gamma, milestones)
            ops.adamw_step(p, grad, m, v, shadow, 0.01, 0.9, 0.99, 1e-15, 0.01, step, zero_grad=True, hyper=hyper,
                           zero_first_n=3072)
            assert int(step_dev) == step and abs(float(hyper[0]) - lr) < 1e-9
            assert float(grad[:3072].abs().max()) == 0.0 and torch.equal(grad[3072:], g[3072:])  # partial re-zeroing
        else:
            ops.adamw_step(p, grad, m, v, shadow, lr, 0.9, 0.99, 1e-15, 0.01, step, zero_grad=True)
            assert float(grad.abs().max()) == 0.0
        assert torch.allclose(p, p_ref.detach(), rtol=2e-5, atol=2e-7), (step, float((p - p_ref.detach()).abs().max()))
        st = opt.state[p_ref]
        # torch forms the moments with lerp / addcmul (different rounding order): compare relative to their scale
        assert torch.allclose(m, st["exp_avg"], rtol=1e-5, atol=1e-6 * float(m.abs().max()))
        assert torch.allclose(v, st["exp_avg_sq"], rtol=1e-5, atol=1e-6 * float(v.abs().max()))
        assert torch.equal(shadow, p.half())  # the fp16 image the kernels read is exactly the rounded parameter


def test_adamw_grad_unscale():
    from nsr_hip import ops
    torch.manual_seed(1)
    n = 4096
    p0 = torch.randn(n, device="cuda")
    g = torch.randn(n, device="cuda")
    outs = []
    for scale in (1.0, 1024.0):
        p, m, v = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        grad = g * scale
        ops.adamw_step(p, grad, m, v, None, 0.01, 0.9, 0.99, 1e-15, 0.01, 1, grad_unscale=1.0 / scale, zero_grad=False)
        assert torch.equal(grad, g * scale)
        outs.append(p)
    assert torch.allclose(outs[0], outs[1], rtol=1e-6, atol=1e-7)


def _nerf_model():
    import nsr
    import refmirror
    torch.manual_seed(0)
    cfg = nsr.configs.get("nerf-blender")
    m = refmirror.NeRFModel(cfg).cuda().train()
    with torch.no_grad():
        m.geometry.encoding_with_network.params[3072:].normal_(0, 0.1)
    ii = torch.stack(torch.meshgrid(*[torch.arange(128)] * 3, indexing="ij"), -1).float().cuda()
    m.occupancy_grid._binary = (((ii + 0.5) / 128 * 3 - 1.5).norm(dim=-1) < 0.9)
    m.randomized = False
    m.background_color = torch.tensor([0.2, 0.4, 0.6], device="cuda")
    g = torch.Generator().manual_seed(3)
    o = torch.nn.functional.normalize(torch.randn(512, 3, generator=g), dim=-1) * 4.0
    d = torch.nn.functional.normalize(-o + torch.randn(512, 3, generator=g) * 0.4, dim=-1)
    return m, torch.cat([o, d], -1).cuda(), torch.rand(512, 3, generator=g).cuda()


def _loss(m, rays, gt):
    out = m(rays)
    valid = out["rays_valid"][..., 0]
    return torch.nn.functional.smooth_l1_loss(out["comp_rgb"][valid], gt[valid])  # systems/nerf.py:97


def test_amp_gradscaler_protocol_matches_fp32_run():
    """Lightning precision 16 = autocast(fp16) around the forward + GradScaler(65536) around backward / step.  The fused
    MLP backward re-scales by its own loss_scale (128) and rounds to fp16 like tcnn, so the first scales may overflow:
    the scaler must SEE that (inf/NaN in .grad -> step skipped, scale halved) and, once a step goes through, the
    unscaled gradients must equal the plain fp32-loss run."""
    import tinycudann as tcnn
    m, rays, gt = _nerf_model()
    mods = [x for x in m.modules() if isinstance(x, tcnn.Module)]
    for x in mods:  # reference gradients: the un-scaled loss with fp32 hand-over between the modules (an un-scaled fp16
        x.dtype = torch.float32  # hand-over underflows for low-weight samples -- the reason Lightning scales the loss)
    _loss(m, rays, gt).backward()
    ref = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None and p.numel()}
    for x in mods:
        x.dtype = torch.float16  # what tcnn hands out, and what the AMP run below uses
    opt = torch.optim.AdamW(m.parameters(), lr=0.0)  # lr 0: the parameters stay put, only the protocol runs
    scaler = torch.amp.GradScaler("cuda", init_scale=65536.0)
    went_through, skipped = 0, 0
    for it in range(16):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            loss = _loss(m, rays, gt)
        assert loss.dtype == torch.float32
        scaler.scale(loss).backward()
        scale = scaler.get_scale()
        scaler.unscale_(opt)
        finite = all(bool(torch.isfinite(p.grad).all()) for p in m.parameters() if p.grad is not None)
        scaler.step(opt)
        scaler.update()
        if not finite:
            skipped += 1
            assert scaler.get_scale() == scale * 0.5  # the overflow was seen: back-off
            continue
        went_through += 1
        for k, p in m.named_parameters():
            if k in ref:
                cos = torch.nn.functional.cosine_similarity(p.grad.flatten(), ref[k].flatten(), dim=0)
                rel = (p.grad - ref[k]).norm() / ref[k].norm()
                assert cos > 0.999 and rel < 3e-2, (it, k, float(cos), float(rel), scale)
        if went_through >= 2:
            break
    assert went_through >= 2, (went_through, skipped)


def test_nonfinite_gradients_reach_the_table_gradient():
    """an inf arriving at the encoding's gradient must surface as inf/NaN in the table gradient (owner-computes path),
    not as a clamped finite value: that is what makes GradScaler skip the step"""
    import tinycudann as tcnn
    from conftest import NERF_GRID
    enc = tcnn.Encoding(3, NERF_GRID).cuda()
    x = torch.rand(5000, 3, device="cuda")
    y = enc(x)
    gy = torch.randn_like(y)
    y.backward(gy, retain_graph=True)
    assert bool(torch.isfinite(enc.params.grad).all())
    enc.params.grad = None
    gy[123, 5] = float("inf")
    gy[4000, 30] = float("nan")
    y.backward(gy)
    g = enc.params.grad
    assert not bool(torch.isfinite(g).all())
    off = [int(v) * 2 for v in enc.grid_desc.offset[:17]]
    bad_levels = {l for l in range(16) if not bool(torch.isfinite(g[off[l]:off[l + 1]]).all())}
    assert bad_levels == {2, 15}, bad_levels  # columns 5 and 30 belong to levels 2 and 15; the others stay clean


def test_scheduled_two_tensor_adamw_is_bit_identical_to_tick_plus_steps():
    """nsr_adamw_step_scheduled (one launch: device schedule + both tensors) == nsr_adam_tick + two nsr_adamw_step"""
    from nsr_hip import ops
    torch.manual_seed(1)
    sizes = (3072 + 50003, 7168)
    milestones, gamma = (3, 6, 8), 0.33

    def state():
        ts = []
        for n in sizes:
            p = torch.randn(n, device="cuda") * 0.1
            ts.append([p, torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(p),
                       torch.empty(n, dtype=torch.float16, device="cuda")])
        return ts, torch.zeros(1, dtype=torch.int32, device="cuda"), torch.zeros(12, dtype=torch.float32, device="cuda")

    torch.manual_seed(1)
    a, step_a, hyper_a = state()
    torch.manual_seed(1)
    b, step_b, hyper_b = state()
    for step in range(1, 12):
        grads = [torch.randn(n, device="cuda") * (10.0 ** torch.randint(-6, 1, (n,), device="cuda").float()) for n in sizes]
        for t, g in zip(a, grads):
            t[1].copy_(g)
        for t, g in zip(b, grads):
            t[1].copy_(g)
        ops.adam_tick(step_a, hyper_a, 0.01, 0.9, 0.99, gamma, milestones)
        ops.adamw_step(a[0][0], a[0][1], a[0][2], a[0][3], a[0][4], 0.01, 0.9, 0.99, 1e-15, 0.01, step, zero_grad=True,
                       hyper=hyper_a, zero_first_n=3072)
        ops.adamw_step(a[1][0], a[1][1], a[1][2], a[1][3], a[1][4], 0.01, 0.9, 0.99, 1e-15, 0.01, step, zero_grad=True,
                       hyper=hyper_a)
        ops.adamw_step_scheduled([tuple(b[0]) + (3072,), tuple(b[1]) + (0,)], step_b, hyper_b, 0.01, 0.9, 0.99, gamma,
                                 milestones, 1e-15, 0.01)
        assert int(step_b) == int(step_a) == step
        assert torch.equal(hyper_a[:8], hyper_b[:8]) and float(hyper_b[8]) == 0.0
        for ta, tb in zip(a, b):
            for x, y in zip(ta, tb):
                assert torch.equal(x, y), step


def test_small_adamw_matches_torch_param_groups():
    """nsr.fused_neus.SmallAdamW (ONE launch over all small tensors, lr per tensor) == torch.optim.AdamW with the NeuS
    YAML's parameter groups (configs/neus-blender.yaml optimizer.params: heads 0.01, variance 0.001), scheduler scale
    applied per step, tensors without a gradient skipped"""
    from nsr.fused_neus import SmallAdamW
    torch.manual_seed(0)
    shapes = [(64, 35), (64, 1), (64,), (13, 64), (13, 1), (13,), ()]
    mine = [torch.nn.Parameter(torch.randn(s, device="cuda") * 0.3) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    opt = SmallAdamW([(p, 0.01) for p in mine[:-1]] + [(mine[-1], 0.001)])
    topt = torch.optim.AdamW([{"params": ref[:-1], "lr": 0.01}, {"params": ref[-1:], "lr": 0.001}], betas=(0.9, 0.99), eps=1e-15)
    for step in range(8):
        scale = 0.01 + 0.99 * step / 7
        for i, (p, r) in enumerate(zip(mine, ref)):
            if step == 3 and i == 1:
                p.grad = r.grad = None  # torch skips a tensor without a gradient (its step counter does not advance either)
                continue
            g = torch.randn(shapes[i], device="cuda") * 10.0 ** float(torch.randint(-5, 1, (1,)))
            p.grad, r.grad = g.clone(), g.clone()
        for grp, base in zip(topt.param_groups, (0.01, 0.001)):
            grp["lr"] = base * scale
        topt.step()
        opt.step(lr_scale=scale)
        assert all(p.grad is None for p in mine)
        for i, (p, r) in enumerate(zip(mine, ref)):
            if i == 1 and step >= 3:
                continue  # bias correction of the skipped tensor lags one step in torch: compared up to the skip only
            assert torch.allclose(p, r, rtol=3e-5, atol=3e-7), (step, i, float((p - r).abs().max()))


def test_adamw_kernels_skip_nonfinite_gradient_elements():
    """the fused trainers have no GradScaler: an inf / NaN gradient element leaves its parameter and moments untouched
    (skip-on-overflow at element granularity) instead of poisoning it; finite elements update as usual"""
    from nsr_hip import ops
    torch.manual_seed(0)
    n = 4099
    p = torch.randn(n, device="cuda")
    m, v = torch.rand(n, device="cuda") * 0.1, torch.rand(n, device="cuda") * 0.01
    g = torch.randn(n, device="cuda")
    g[5], g[77], g[4098] = float("inf"), float("nan"), float("-inf")
    bad = torch.tensor([5, 77, 4098], device="cuda")
    p0, m0, v0 = p.clone(), m.clone(), v.clone()
    shadow = torch.empty(n, dtype=torch.float16, device="cuda")
    ops.adamw_step(p, g.clone(), m, v, shadow, 0.01, 0.9, 0.99, 1e-15, 0.01, 3, zero_grad=True)
    assert bool(torch.isfinite(p).all()) and bool(torch.isfinite(m).all()) and bool(torch.isfinite(v).all())
    assert torch.equal(p[bad], p0[bad]) and torch.equal(m[bad], m0[bad]) and torch.equal(v[bad], v0[bad])
    good = torch.ones(n, dtype=torch.bool, device="cuda")
    good[bad] = False
    assert bool((p[good] != p0[good]).all())
