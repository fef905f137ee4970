"""Host-side logic of the product that needs no GPU (-m "not gpu")."""
def test_sorted_uniform_is_the_order_statistics_of_iid_uniforms():
    """nsr.fused_neus.sorted_uniform_ (the random cells of a grid refresh, drawn in increasing order): sorted, inside [0, 1),
    and distributed like the sorted values of i.i.d. uniforms -- the k-th of n has mean k / (n + 1)"""
    import torch
    from nsr.fused_neus import sorted_uniform_
    torch.manual_seed(3)
    n = 20000
    u = sorted_uniform_(torch.empty(n))
    assert bool((u[1:] >= u[:-1]).all()) and float(u[0]) >= 0.0 and float(u[-1]) < 1.0
    k = torch.arange(1, n + 1, dtype=torch.float64)
    # (std of the k-th order statistic <= 0.5 / sqrt(n): 6 sigma)
    assert float((u.double() - k / (n + 1)).abs().max()) < 6 * 0.5 / n ** 0.5
    cells = (u * 4096).long().clamp_(max=4095)  # what the refresh does with them: cell indices, uniformly hit
    hist = torch.bincount(cells, minlength=4096).double()
    assert abs(float(hist.mean()) - n / 4096) < 1e-9 and float(hist.max()) < 30
    assert sorted_uniform_(torch.empty(0)).numel() == 0


def _torch_schedule(make, n):
    """learning-rate factor per optimizer step of a torch scheduler stack on a unit learning rate"""
    import torch
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1.0)
    sched = make(opt)
    out = []
    for _ in range(n):
        out.append(opt.param_groups[0]["lr"])
        opt.step()
        sched.step()
    return out


def test_lr_schedules_match_the_torch_schedulers_of_the_reference_yamls():
    """nsr.trainer.multistep_lr_scale = MultiStepLR of configs/nerf-blender.yaml:80-85; nsr.fused_neus.neus_lr_scale =
    SequentialLR(LinearLR 0.01 -> 1 over 500 steps, ExponentialLR) of configs/neus-blender.yaml and the constant-then-
    exponential schedule of neus-dtu / neuralangelo (systems/utils.py builds them with interval: step)"""
    import torch
    from torch.optim import lr_scheduler as S
    from nsr.fused_neus import neus_lr_scale
    from nsr.trainer import multistep_lr_scale
    n, T = 20000, 20000
    want = _torch_schedule(lambda o: S.MultiStepLR(o, milestones=[10000, 15000, 18000], gamma=0.33), n)
    for t in (0, 9999, 10000, 14999, 15000, 17999, 18000, 19999):
        assert abs(multistep_lr_scale(t) - want[t]) < 1e-12, t
    want = _torch_schedule(lambda o: S.SequentialLR(o, [S.LinearLR(o, start_factor=0.01, end_factor=1.0, total_iters=500),
                                                        S.ExponentialLR(o, gamma=0.1 ** (1.0 / (T - 500)))],
                                                    milestones=[500]), n)
    for t in (0, 1, 250, 499, 500, 501, 5000, 19999):
        assert abs(neus_lr_scale(t, "neus-blender", T) - want[t]) < 1e-9 * max(want[t], 1e-3) + 1e-12, (t, want[t])
    want = _torch_schedule(lambda o: S.SequentialLR(o, [S.ConstantLR(o, factor=1.0, total_iters=5000),
                                                        S.ExponentialLR(o, gamma=0.1 ** (1.0 / (T - 5000)))],
                                                    milestones=[5000]), n)
    for name in ("neus-dtu", "neuralangelo"):
        for t in (0, 4999, 5000, 5001, 12000, 19999):
            assert abs(neus_lr_scale(t, name, T) - want[t]) < 1e-9, (name, t, want[t])


def test_sample_buffer_capacity_controller():
    """nsr.trainer.next_capacity (lagged statistics, no host wait): grows as soon as the window maximum passes 85 % of the
    capacity, shrinks only when the buffers are more than twice too large and nothing was dropped, stays otherwise"""
    from nsr.trainer import next_capacity
    cap = 262144
    assert next_capacity(cap, 0, 8192, 8192, False) == cap                      # no statistics yet
    grown = next_capacity(cap, int(0.9 * cap), 8192, 8192, False)
    assert grown > cap and grown % 16384 == 0
    assert next_capacity(cap, int(0.5 * cap), 8192, 8192, False) == cap          # comfortable: unchanged
    assert next_capacity(cap, int(0.5 * cap), 8192, 8192, True) == cap           # something was dropped: never shrink
    shrunk = next_capacity(cap, 20000, 8192, 8192, False)
    assert 65536 <= shrunk < cap // 2 + 16384
    # few rays in use of many slots: the ray count still has room to climb, and the counts with it
    assert next_capacity(cap, 60000, 1024, 8192, False) >= next_capacity(cap, 60000, 8192, 8192, False)


def test_deferred_selections_and_lazy_outputs_on_the_host():
    """nsr.models._ValidMask / _MaskedRows / _LazyOutputs / _LazyCount (the model entries' non-synchronising outputs) on CPU
    tensors -- the host logic alone: a selection by the validity mask is deferred, the systems' mean losses over two such
    selections (systems/nerf.py:97, systems/neus.py:98,102) equal the losses of the gathered rows in value and gradient (here
    through the elementwise fallback: the kernels need a GPU), anything else gathers, per-sample outputs are sliced when read"""
    import torch
    import torch.nn.functional as F
    from nsr.models import _LazyCount, _LazyOutputs, _MaskedRows, _ValidMask
    g = torch.Generator().manual_seed(0)
    pred = torch.rand(300, 3, generator=g, requires_grad=True)
    target = torch.rand(300, 3, generator=g)
    mask = torch.rand(300, 1, generator=g) < 0.6
    valid = mask.as_subclass(_ValidMask)[..., 0]
    assert type(valid) is _ValidMask and valid.shape == (300,)
    a, b = pred[valid], target[valid]
    assert isinstance(a, _MaskedRows) and isinstance(b, _MaskedRows)
    for fn, kw in ((F.smooth_l1_loss, {}), (F.smooth_l1_loss, {"beta": 0.1}), (F.mse_loss, {}), (F.l1_loss, {}),
                   (F.huber_loss, {"delta": 0.2})):
        ref = fn(pred[mask[:, 0]], target[mask[:, 0]], **kw)
        out = fn(a, b, **kw)
        assert torch.allclose(out, ref, rtol=1e-6, atol=1e-9), fn.__name__
        g_ref, = torch.autograd.grad(ref, pred)
        g_out, = torch.autograd.grad(out, pred)
        assert torch.allclose(g_out, g_ref, rtol=1e-5, atol=1e-9), fn.__name__
    # a reduction the deferred form does not know, and attribute access, gather the rows (the reference's behaviour)
    assert torch.equal(F.mse_loss(a, b, reduction="sum"), F.mse_loss(pred[mask[:, 0]], target[mask[:, 0]], reduction="sum"))
    assert a.shape == (int(mask.sum()), 3) and torch.equal(a.detach(), pred.detach()[mask[:, 0]])
    # operators are looked up on the type: they gather too
    rows, trow = pred.detach()[mask[:, 0]], target[mask[:, 0]]
    assert torch.equal((a - b).detach(), rows - trow) and torch.equal((2.0 * a).detach(), 2.0 * rows) and torch.equal((-a).detach(), -rows)
    assert torch.equal(((a - b).abs().mean()).detach(), (rows - trow).abs().mean()) and torch.equal(abs(b), trow.abs())
    assert len(a) == rows.shape[0] and torch.equal(a[0].detach(), rows[0]) and torch.equal((a > 0.5), rows > 0.5)
    assert torch.equal((a ** 2).detach(), rows ** 2) and torch.equal((a / (b + 1.0)).detach(), rows / (trow + 1.0))
    # selections over DIFFERENT masks are not fused
    other = (~mask).as_subclass(_ValidMask)[..., 0]
    assert torch.equal((pred[valid].materialize()).detach(), pred.detach()[mask[:, 0]])
    assert pred[other].shape[0] + pred[valid].shape[0] == 300
    # the mask itself stays an ordinary bool tensor for everything else
    assert int(valid.sum()) == int(mask.sum()) and type(valid.sum()) is torch.Tensor
    assert type(~valid) is not _MaskedRows and torch.equal((valid | ~valid).as_subclass(torch.Tensor), torch.ones(300, dtype=torch.bool))

    class Handle:  # what FusedNeRFStep.render_forward(lazy=True) hands over: (marched, kept) of the last / this call
        def __init__(self, prev, cur):
            self.prev, self.cur, self.waits = prev, cur, 0

        def previous(self):
            return self.prev

        def current(self):
            self.waits += 1
            return self.cur

    h = Handle((1000, 400), (900, 350))
    n = _LazyCount(h)
    assert n.sum().item() == 400 and int(n) == 400 and n.current() == 350 and h.waits == 1
    assert int(_LazyCount(Handle(None, (7, 5))).sum().item()) == 5           # the first forward has no predecessor
    assert torch.equal(torch.as_tensor([3]) + n, torch.as_tensor([403]))    # any torch function sees a one-element tensor
    assert torch.equal(n + 1, torch.as_tensor([401], dtype=torch.int32)) and bool((n.sum() > 0).all()) and float(n.sum() / 2) == 200.0
    assert n.shape == (1,) and n.dtype == torch.int32 and n.float().item() == 400.0 and bool(n)
    h = Handle((1000, 400), (900, 350))
    w = torch.arange(512.)
    out = _LazyOutputs({"comp_rgb": pred}, {"weights": w, "points": lambda: w * 2.0}, _LazyCount(h))
    assert h.waits == 0 and out["comp_rgb"] is pred and h.waits == 0         # ray outputs: no wait
    assert out["weights"].shape == (350,) and h.waits == 1                   # sample outputs: sliced to the live count when read
    assert torch.equal(out.get("points"), w[:350] * 2.0) and set(out) == {"comp_rgb", "weights", "points"}
    assert all(v is not None for v in dict(out.items()).values())
