"""The fused step BEHIND the reference's model interface (nsr/models.py): ``models.make('nerf', cfg)`` -> FusedNeRFModel.

The statements of the reference's training step (systems/nerf.py:87-106: ``out = self.model(rays)``, dynamic ray count from
``out['num_samples']``, smooth-L1 on the valid rays, distortion loss on weights / points / intervals, ``backward()``) are run
on the fused entry and on the reference's own model restated on the drop-in packages (tests/refmirror, the modular autograd
path); outputs and parameter gradients have to agree, and with the plain loss the gradients equal the fused trainer's."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu


def _pair(seed=0):
    """the reference's model (modular path) and the fused entry with the SAME parameters and occupancy grid"""
    import nsr
    import nsr.models
    import refmirror
    import tinycudann as tcnn
    torch.manual_seed(seed)
    cfg = nsr.configs.get("nerf-blender")
    ref = refmirror.NeRFModel(cfg).cuda().train()
    with torch.no_grad():
        ref.geometry.encoding_with_network.params[3072:].normal_(0, 0.08)
    g = ref.occupancy_grid
    ii = torch.stack(torch.meshgrid(*[torch.arange(128)] * 3, indexing="ij"), -1).float().cuda()
    g._binary = (((ii + 0.5) / 128 * 3 - 1.5).norm(dim=-1) < 1.1)
    for m in ref.modules():
        if isinstance(m, tcnn.Module):
            m.dtype = torch.float32
    fused = nsr.models.FusedNeRFModel(cfg).cuda().train()
    missing = fused.load_state_dict(ref.state_dict(), strict=True)  # identical key set: the reference's checkpoints load
    assert not missing.missing_keys and not missing.unexpected_keys
    for m in (ref, fused):
        m.randomized = False
    return ref, fused, cfg


def _rays(n, seed=1):
    g = torch.Generator().manual_seed(seed)
    o = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1) * 4.0
    d = torch.nn.functional.normalize(-o + torch.randn(n, 3, generator=g) * 0.5, dim=-1)
    return torch.cat([o, d], -1).cuda(), torch.rand(n, 3, generator=g).cuda()


def _system_step(model, rays, rgb, lambda_distortion, extra):
    """systems/nerf.py:87-106 (+ optional terms on opacity / depth so that every upstream gradient is exercised)"""
    from torch_efficient_distloss import flatten_eff_distloss
    out = model(rays)
    n = int(out["num_samples"].sum().item())
    valid = out["rays_valid"][..., 0]
    loss = torch.nn.functional.smooth_l1_loss(out["comp_rgb"][valid], rgb[valid])
    if lambda_distortion > 0:
        loss = loss + lambda_distortion * flatten_eff_distloss(out["weights"], out["points"], out["intervals"], out["ray_indices"])
    if extra:
        loss = loss + 0.01 * (out["opacity"] ** 2).mean() + 0.003 * out["depth"].mean()
    loss.backward()
    return out, n, float(loss)


def _rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("lambda_distortion,extra", [(0.0, False), (0.01, True)], ids=["smooth_l1", "all_terms"])
def test_fused_entry_runs_the_reference_system_step(lambda_distortion, extra):
    ref, fused, cfg = _pair()
    rays, rgb = _rays(700)
    bg = torch.tensor([0.3, 0.6, 0.9], device="cuda")
    ref.background_color = fused.background_color = bg
    o_ref, n_ref, l_ref = _system_step(ref, rays, rgb, lambda_distortion, extra)
    o_fu, n_fu, l_fu = _system_step(fused, rays, rgb, lambda_distortion, extra)
    assert n_ref == n_fu and n_fu > 10000
    assert set(o_fu) == set(o_ref)                                       # the reference's training output dict
    assert torch.equal(o_fu["ray_indices"], o_ref["ray_indices"])        # bit-exact segment indices
    assert torch.equal(o_fu["points"], o_ref["points"]) and torch.equal(o_fu["intervals"], o_ref["intervals"])
    assert torch.equal(o_fu["rays_valid"], o_ref["rays_valid"]) and o_fu["rays_valid"].dtype == torch.bool
    for k, (rt, at) in {"comp_rgb": (1e-4, 2e-5), "opacity": (1e-4, 1e-5), "depth": (1e-4, 1e-4), "weights": (1e-4, 1e-6)}.items():
        assert o_fu[k].shape == o_ref[k].shape, k
        assert torch.allclose(o_fu[k], o_ref[k], rtol=rt, atol=at), k
    assert abs(l_fu - l_ref) < 1e-5 * max(1.0, abs(l_ref))
    for path in ("geometry.encoding_with_network", "texture.network"):
        a, b = fused.get_submodule(path).params.grad, ref.get_submodule(path).params.grad
        assert a is not None and a.shape == b.shape
        if path.startswith("geometry"):
            assert _rel(a[:3072], b[:3072]) < 5e-3 and _rel(a[3072:], b[3072:]) < 5e-3
        else:
            assert _rel(a, b) < 5e-3
    if lambda_distortion == 0.0 and not extra:
        # ... and the same loss through the fused TRAINER's step (loss formed inside the kernels): same gradients
        from nsr.fused import FusedNeRFStep
        step = FusedNeRFStep(ref)
        ref.zero_grad(set_to_none=True)
        step.forward_backward(rays, rgb, bg)
        for path in ("geometry.encoding_with_network", "texture.network"):
            assert _rel(fused.get_submodule(path).params.grad, ref.get_submodule(path).params.grad) < 2e-5, path


def test_fused_entry_eval_is_chunked_and_gradient_free():
    _, fused, cfg = _pair()
    rays, _ = _rays(300)
    fused.background_color = torch.ones(3, device="cuda")
    train_out = fused(rays)
    fused.eval()
    fused.config["ray_chunk"] = 128  # three chunks
    out = fused(rays)
    assert set(out) == {"comp_rgb", "opacity", "depth", "rays_valid", "num_samples"}
    assert not out["comp_rgb"].requires_grad and out["comp_rgb"].device.type == "cpu"   # chunk_batch(..., move_to_cpu=True)
    assert torch.allclose(out["comp_rgb"], train_out["comp_rgb"].detach().cpu(), atol=1e-6)
    assert int(out["num_samples"].sum()) == int(train_out["num_samples"].sum())
    fused.train()
    assert fused.randomized == bool(cfg["randomized"])


def test_registry_switch_and_a_short_training_run():
    """`nsr.models.register(models)` points the reference's registry at the fused entry; a few hundred steps of the system's
    loop (torch AdamW, the model's own occupancy refresh through update_step) reduce the loss"""
    import nsr
    import nsr.models
    from nsr.scene import SyntheticBlender
    registry = types.SimpleNamespace(models={"nerf": object})
    nsr.models.register(registry)
    torch.manual_seed(0)
    cfg = nsr.configs.get("nerf-blender")
    model = registry.models["nerf"](cfg).cuda().train()
    data = SyntheticBlender(n_images=8, w=100, h=100, device="cuda", seed=0)
    gen = torch.Generator(device="cuda").manual_seed(3)
    opt = torch.optim.AdamW(model.parameters(), lr=0.01, betas=(0.9, 0.99), eps=1e-15)
    n_rays, losses = 256, []
    for step in range(330):
        rays, rgb, fg, bg = data.sample_rays(n_rays, gen, "random")
        model.background_color = bg
        model.update_step(0, step)
        out = model(rays)
        n = int(out["num_samples"].sum().item())
        if n > 0:
            n_rays = min(int(n_rays * 0.9 + int(n_rays * (256 * 1024 / n)) * 0.1), 8192)
        valid = out["rays_valid"][..., 0]
        loss = torch.nn.functional.smooth_l1_loss(out["comp_rgb"][valid], rgb[valid])
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert sum(losses[-5:]) / 5 < 0.5 * sum(losses[:5]) / 5, (losses[:5], losses[-5:])
    assert float(model.occupancy_grid.binary.float().mean()) < 0.9   # the model's update_step pruned the grid


# ---- models.make('neus', cfg) -> FusedNeuSModel --------------------------------------------------------------------------
NEUS_LAMBDAS = {"lambda_rgb_mse": 10.0, "lambda_rgb_l1": 1.0, "lambda_mask": 0.1, "lambda_opaque": 0.05, "lambda_eikonal": 0.1,
                "lambda_sparsity": 0.02, "sparsity_scale": 1.0}


def _neus_pair(name, step):
    """the reference's NeuS model (modular path, tests/refmirror) and the fused entry with the same parameters / grids"""
    import nsr
    import nsr.models
    import refmirror
    torch.manual_seed(0)
    cfg = nsr.configs.get(name)
    cfg["num_samples_per_ray"] = 256
    ref = refmirror.NeuSModel(cfg).cuda().train()
    with torch.no_grad():
        enc = ref.geometry.encoding.encoding
        (enc.encoding if hasattr(enc, "encoding") else enc).params.normal_(0, 0.05)
        ref.geometry.network.layers[0].weight_v[:, 3:].normal_(0, 0.05)
        if cfg.get("learned_background"):
            ref.geometry_bg.encoding_with_network.encoding.encoding.params.normal_(0, 0.3)
    r = float(cfg["radius"])
    ii = torch.stack(torch.meshgrid(*[torch.arange(128)] * 3, indexing="ij"), -1).float().cuda()
    ref.occupancy_grid._binary = (((ii + 0.5) / 128 * 2 * r - r).norm(dim=-1) < 0.55 * r)
    if cfg.get("learned_background"):
        jj = torch.stack(torch.meshgrid(*[torch.arange(256)] * 3, indexing="ij"), -1).cuda()
        ref.occupancy_grid_bg._binary = ((jj.sum(-1) % 3) != 0)  # a deterministic 2/3-full pattern of the contracted space
    fused = nsr.models.FusedNeuSModel(cfg).cuda().train()
    res = fused.load_state_dict(ref.state_dict(), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    for m in (ref, fused):
        m.update_step(0, step)   # not a multiple of 16: schedules only, no occupancy refresh
        m.randomized = False
        m.background_color = torch.tensor([0.3, 0.5, 0.7], device="cuda")
    return ref, fused, cfg


@pytest.mark.parametrize("name,step", [("neus-blender", 7001), ("neuralangelo", 9005), ("neus-dtu", 7001)])
def test_fused_neus_entry_runs_the_reference_system_step(name, step):
    """systems/neus.py:88-139 on both models: same output dict, same losses, same parameter gradients"""
    import fixture_utils as fu
    from torch_efficient_distloss import flatten_eff_distloss
    ref, fused, cfg = _neus_pair(name, step)
    g = torch.Generator().manual_seed(1)
    scale = float(cfg["radius"]) / 1.5
    o = torch.nn.functional.normalize(torch.randn(300, 3, generator=g), dim=-1) * 4.0 * scale
    d = torch.nn.functional.normalize(-o + torch.randn(300, 3, generator=g) * 0.45 * scale, dim=-1)
    rays = torch.cat([o, d], -1).cuda()
    rgb = torch.rand(300, 3, generator=g).cuda()
    fg = (torch.rand(300, generator=g) > 0.4).float().cuda()
    lam = dict(NEUS_LAMBDAS, lambda_curvature=(1e-4 if name == "neuralangelo" else 0.0))
    outs, grads = [], []
    for m in (ref, fused):
        out = m(rays)
        loss, terms = fu.neus_system_loss(out, rgb, fg, lam)
        loss = loss + 0.01 * flatten_eff_distloss(out["weights"], out["points"], out["intervals"], out["ray_indices"])
        loss.backward()
        outs.append((out, float(loss), {k: float(v) for k, v in terms.items()}))
        grads.append({k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None and p.numel()})
    (o_ref, l_ref, t_ref), (o_fu, l_fu, t_fu) = outs
    assert set(o_fu) == set(o_ref), set(o_fu) ^ set(o_ref)
    assert int(o_fu["num_samples"]) == int(o_ref["num_samples"]) > 5000
    assert int(o_fu["num_samples_full"]) == int(o_ref["num_samples_full"])
    assert torch.equal(o_fu["ray_indices"], o_ref["ray_indices"])
    assert torch.equal(o_fu["rays_valid_full"], o_ref["rays_valid_full"])
    for k in ("comp_rgb_full", "comp_rgb", "opacity", "depth", "comp_normal", "sdf_samples", "sdf_grad_samples", "weights",
              "points", "intervals"):
        a, b = o_fu[k].detach().reshape(-1), o_ref[k].detach().reshape(-1)
        assert a.shape == b.shape, k
        bad = float(((a - b).abs() > 2e-3 + 2e-3 * b.abs()).float().mean())  # (cell-face outliers: see test_gpu_fused_neus.py)
        assert bad < 5e-3, (name, k, bad, float((a - b).abs().max()))
    assert abs(float(o_fu["inv_s"]) - float(o_ref["inv_s"])) < 1e-5 * float(o_ref["inv_s"])
    for k in t_ref:
        assert abs(t_fu[k] - t_ref[k]) <= 2e-3 * abs(t_ref[k]) + 1e-5, (name, k, t_fu[k], t_ref[k])
    assert abs(l_fu - l_ref) < 2e-3 * abs(l_ref)
    assert set(grads[1]) == set(grads[0]), set(grads[1]) ^ set(grads[0])
    for k, w in grads[0].items():
        fu.assert_grad(grads[1][k], w, (name, k))


def test_fused_neus_entry_accumulates_gradients_and_evaluates_in_chunks():
    """two backward passes before an optimizer step (accumulate_grad_batches) add up; eval is chunked and gradient-free"""
    _, fused, cfg = _neus_pair("neus-blender", 7001)
    g = torch.Generator().manual_seed(2)
    o = torch.nn.functional.normalize(torch.randn(128, 3, generator=g), dim=-1) * 4.0
    rays = torch.cat([o, torch.nn.functional.normalize(-o, dim=-1)], -1).cuda()

    def one():
        out = fused(rays)
        (out["comp_rgb_full"].mean() + 0.1 * ((out["sdf_grad_samples"].norm(dim=-1) - 1) ** 2).mean()).backward()

    one()
    first = {k: p.grad.clone() for k, p in fused.named_parameters() if p.grad is not None}
    one()
    for k, p in fused.named_parameters():
        if k in first:
            assert fu_rel(p.grad, 2 * first[k]) < 1e-5, k
    fused.eval()
    fused.config["ray_chunk"] = 50
    out = fused(rays)
    assert out["comp_rgb_full"].device.type == "cpu" and not out["comp_rgb_full"].requires_grad
    assert "sdf_samples" not in out and out["comp_rgb_full"].shape == (128, 3)


def fu_rel(a, b):
    import fixture_utils as fu
    return fu.rel_l2(a, b)
