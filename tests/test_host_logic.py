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
