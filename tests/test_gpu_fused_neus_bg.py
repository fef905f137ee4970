"""The fused NeuS step with the NeRF++ background (configs/neus-dtu.yaml, BASELINE config C4; reference
models/neus.py:141-203 `forward_bg_` + 259-287) against (a) the fixture the REFERENCE's own models/ produced
(neus_bg_forward.npz, tests/gen_golden.py:gen_neus_bg) and (b) the modular drop-in path on the same full-size model
with every loss term switched on.  Tolerances as in test_gpu_golden.py / test_gpu_fused_neus.py."""
import numpy as np
import pytest
import torch

import fixture_utils as fu
from test_golden_glue import SMALL_GRID, binary_from, load

pytestmark = pytest.mark.gpu

BG_KEYS = ("geometry_bg.encoding_with_network.encoding.encoding.params",
           "geometry_bg.encoding_with_network.network.layers.0.weight",
           "geometry_bg.encoding_with_network.network.layers.0.bias",
           "geometry_bg.encoding_with_network.network.layers.2.weight",
           "texture_bg.network.layers.0.weight", "texture_bg.network.layers.2.weight", "texture_bg.network.layers.4.weight",
           "texture_bg.network.layers.4.bias")


def _cos(a, b):
    return float(torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0))


def test_fused_neus_background_matches_reference_fixture():
    import nsr
    import refmirror
    from nsr.fused_neus import FusedNeuSStep
    fx = load("neus_bg_forward.npz")
    cfg = nsr.configs.get("neus-dtu")
    for key in ("geometry", "geometry_bg"):
        cfg[key]["xyz_encoding_config"].update({k: v for k, v in SMALL_GRID.items() if k != "otype"})
    cfg["num_samples_per_ray"] = 256
    m = refmirror.NeuSModel(cfg).cuda().train()
    sd = {k[len("param/"):]: v for k, v in fx.items() if k.startswith("param/")}
    m.load_state_dict(sd, strict=False)
    m.update_step(0, 5000)
    m.occupancy_grid._binary = binary_from(fx).cuda()
    m.occupancy_grid_bg._binary = torch.from_numpy(np.unpackbits(fx["binary_bg_packed"].numpy())[:256 ** 3]
                                                   .reshape(256, 256, 256).astype(bool)).cuda()
    m.background_color = fx["background"].cuda()
    m.randomized = False
    rays = fx["rays"].cuda()
    # the fixture's loss: mse(comp_rgb_full, 0.4) * 10 + eikonal * 0.1 over every ray (all of them are valid here)
    step = FusedNeuSStep(m, dict(lambda_rgb_l1=0.0, lambda_rgb_mse=10.0, lambda_eikonal=0.1, lambda_mask=0.0))
    gt = torch.full((rays.shape[0], 3), 0.4, device="cuda")
    res = step.forward_backward(rays, gt, None, m.background_color)
    assert torch.equal(res["ray_indices"].cpu(), fx["out/ray_indices"])
    assert torch.equal(res["ray_indices_bg"].cpu(), fx["out/ray_indices_bg"])  # same marching + pruning decisions
    mid = (res["t_starts_bg"] + res["t_ends_bg"]) / 2
    assert torch.allclose(mid.cpu(), fx["out/points_bg"].view(-1), rtol=1e-6, atol=1e-6)
    for k in ("comp_rgb_bg", "opacity_bg", "depth_bg", "comp_rgb", "opacity", "comp_rgb_full", "weights_bg"):
        tol = 2e-2 if k == "depth_bg" else 3e-3
        want = fx["out/" + k]
        assert torch.allclose(res[k].cpu().view(want.shape), want, atol=tol, rtol=1e-2), \
            (k, float((res[k].cpu().view(want.shape) - want).abs().max()))
    assert torch.equal(res["rays_valid_full"].cpu().view(-1), fx["out/rays_valid_full"].view(-1))
    assert res["num_samples_full"] == int(fx["out/num_samples_full"])
    assert bool(res["rays_valid_full"].all())
    assert abs(float(step.loss_value(res["loss_acc"])) - float(fx["loss"])) < 3e-3 * max(1.0, float(fx["loss"]))
    params = dict(m.named_parameters())
    for k in BG_KEYS + ("geometry.encoding.encoding.params", "texture.network.layers.0.weight",
                        "geometry.network.layers.0.weight_v", "geometry.network.layers.2.weight_v"):
        assert params[k].grad is not None, k
        fu.assert_grad(params[k].grad, fx["grad/" + k], k)


def test_fused_neus_background_matches_modular_path():
    """full-size neus-dtu model, same rays: FusedNeuSStep vs autograd over the drop-in packages, all loss terms on"""
    import nsr
    import refmirror
    from nsr.fused_neus import FusedNeuSStep
    torch.manual_seed(0)
    cfg = nsr.configs.get("neus-dtu")
    cfg["num_samples_per_ray"] = 256
    m = refmirror.NeuSModel(cfg).cuda().train()
    with torch.no_grad():
        m.geometry.encoding.encoding.params.normal_(0, 0.05)
        m.geometry.network.layers[0].weight_v[:, 3:].normal_(0, 0.05)
        m.geometry_bg.encoding_with_network.encoding.encoding.params.normal_(0, 0.3)
    m.update_step(0, 7001)
    ii = torch.stack(torch.meshgrid(*[torch.arange(128)] * 3, indexing="ij"), -1).float().cuda()
    m.occupancy_grid._binary = (((ii + 0.5) / 128 * 2 - 1.0).norm(dim=-1) < 0.6)
    jj = torch.stack(torch.meshgrid(*[torch.arange(256)] * 3, indexing="ij"), -1).cuda()
    m.occupancy_grid_bg._binary = ((jj.sum(-1) % 3) != 0)
    m.background_color = torch.tensor([0.3, 0.5, 0.7], device="cuda")
    m.randomized = False
    g = torch.Generator().manual_seed(1)
    o = torch.nn.functional.normalize(torch.randn(300, 3, generator=g), dim=-1) * 2.4
    d = torch.nn.functional.normalize(-o + torch.randn(300, 3, generator=g) * 0.8, dim=-1)  # some rays miss the box
    rays = torch.cat([o, d], -1).cuda()
    gt = torch.rand(300, 3, generator=g).cuda()
    fg = (torch.rand(300, generator=g) > 0.4).float().cuda()
    lam = {"lambda_rgb_l1": 1.0, "lambda_rgb_mse": 0.5, "lambda_mask": 0.1, "lambda_opaque": 0.05, "lambda_eikonal": 0.1,
           "lambda_sparsity": 0.02, "sparsity_scale": 1.0}
    out = m(rays)
    loss, terms = fu.neus_system_loss(out, gt, fg, lam)
    loss.backward()
    ref = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None and p.numel()}
    for p in m.parameters():
        p.grad = None
    step = FusedNeuSStep(m, lam)
    res = step.forward_backward(rays, gt, fg, m.background_color)
    assert res["num_samples"] == int(out["num_samples"])
    assert res["num_samples_bg"] == int(out["num_samples_bg"]) and res["num_samples_bg"] > 1000
    assert torch.equal(res["ray_indices_bg"], out["ray_indices_bg"])
    assert torch.equal(res["rays_valid_full"].view(-1), out["rays_valid_full"].view(-1))
    for k in ("comp_rgb_bg", "opacity_bg", "comp_rgb_full", "opacity", "weights_bg"):
        a, b = res[k].reshape(-1), out[k].detach().reshape(-1)
        bad = float(((a - b).abs() > 2e-3 + 2e-3 * b.abs()).float().mean())
        assert bad < 5e-3, (k, float((a - b).abs().max()), bad)
    mine = step.loss_terms(res["loss_acc"])
    for k in ("rgb_l1", "rgb_mse", "mask", "opaque", "eikonal", "sparsity"):
        assert abs(float(mine[k]) - float(terms[k])) <= 2e-3 * abs(float(terms[k])) + 1e-5, (k, float(mine[k]), float(terms[k]))
    assert abs(float(step.loss_value(res["loss_acc"])) - float(loss)) < 2e-3 * abs(float(loss))
    now = dict(m.named_parameters())
    for k, w in ref.items():
        assert now[k].grad is not None, k
        fu.assert_grad(now[k].grad, w, k)


def test_neus_dtu_trainer_steps_with_background():
    """NeuSTrainer on the product state holder: the background branch trains (its parameters move, its 256^3 grid is
    refreshed), the sample budget counts both branches (systems/neus.py:93-95)"""
    import nsr
    from nsr.fused_neus import NeuSTrainer
    from nsr.scene import SyntheticBlender
    torch.manual_seed(0)
    cfg = nsr.configs.get("neus-dtu")
    st = nsr.build(cfg).cuda().train()
    ds = SyntheticBlender(n_images=8, h=64, w=64, device="cuda", environment=True)
    tr = NeuSTrainer(st, ds, cfg, {"lambda_rgb_l1": 1.0, "lambda_eikonal": 0.1}, config_name="neus-dtu")
    before = {k: v.detach().clone() for k, v in st.named_parameters() if "_bg" in k and v.numel()}
    for _ in range(20):
        last = tr.train_step()
    assert last["n_samples"] > 0 and last["n_samples_bg"] > 0
    acc = last["loss_acc"]
    assert bool(torch.isfinite(acc).all())
    moved = [k for k, v in st.named_parameters() if k in before and not torch.equal(v.detach(), before[k])]
    assert len(moved) == len(before), set(before) - set(moved)
    assert 0 < int(st.occupancy_grid_bg.binary.sum()) <= 256 ** 3


@pytest.mark.parametrize("name", ["neus-blender", "neus-dtu", "neuralangelo"])
def test_table_adamw_inside_the_backward_matches_the_separate_sweep(name):
    """one GPU: NeuSTrainer hands AdamW on the hash tables to the owner-computes backward (analytic mode with the second-order
    term, the NeRF++ background's table, the finite-difference stencil mode).  Same parameters as gradient store + k_adamw:
    the first step bit for bit on the hashed levels (fixed-point sums) up to the bias-correction rounding (device-side
    running beta powers vs the host's pow); later steps within 15 % of one step's movement in norm."""
    import nsr
    from nsr.fused_neus import NeuSTrainer
    from nsr.scene import SyntheticBlender
    cfg = nsr.configs.get(name)
    runs = {}
    for fused in (True, False):
        torch.manual_seed(0)
        st = nsr.build(cfg).cuda().train()
        ds = SyntheticBlender(n_images=8, h=64, w=64, device="cuda", environment=bool(cfg["learned_background"]), seed=0)
        tr = NeuSTrainer(st, ds, cfg, {"lambda_rgb_l1": 1.0, "lambda_eikonal": 0.1}, config_name=name)
        tr.fuse_table_adam = fused
        if name == "neuralangelo":
            tr.global_step = 12000
        t0 = tr.global_step
        snaps = []
        for _ in range(3):
            last = tr.train_step()
            assert last["n_samples"] > 0
            assert (tr.fused.adam_applied == set(tr._table_of)) == fused
            snaps.append({k: m.params.detach().clone() for k, m in tr._table_of.items()})
        runs[fused] = snaps
        assert tr.global_step == t0 + 3
    lr = 0.01
    for step in range(3):
        for k in runs[True][step]:
            a, b = runs[True][step][k], runs[False][step][k]
            if step == 0:  # AdamW moves an entry by at most ~lr per step: compare on that scale
                # (a few entries of the small dense levels, whose slabs are summed in fp32 in a different order, have a
                # gradient that is pure cancellation noise: with eps = 1e-15 its SIGN decides a full +-lr step)
                far = ((a - b).abs() > 2e-6 * lr + 1e-9).float().mean()
                assert float(far) < 1e-3, (name, step, k, float(far))
            else:
                # with eps = 1e-15 the normalised step of an entry whose gradient is rounding noise is +-lr whatever its size:
                # trajectories of such entries separate, so later steps are compared in norm, against one step's movement
                moved = float((a - runs[True][step - 1][k]).norm())
                assert float((a - b).norm()) <= 0.15 * moved, (name, step, k, float((a - b).norm()), moved)
            assert not torch.equal(a, torch.zeros_like(a))


@pytest.mark.parametrize("step", [0, 512])
def test_device_background_refresh_matches_torch_update(step):
    """FusedNeuSStep.refresh_bg_occupancy_async (cell selection, tile-major encode, density head on the encoding alone,
    exp(logit + bias) * step inside the unit sphere, EMA / threshold / binarise, bricks) == nerfacc's
    OccupancyGrid._update_cells with the reference's background occ_eval_fn (models/neus.py:103-106) on the cells / jitter
    the kernels selected: samples outside the unit sphere are dropped by the selection, their cells keep their value"""
    import copy
    import nsr
    from nsr.fused_neus import FusedNeuSStep
    from nsr_hip import ops
    torch.manual_seed(0)
    cfg = nsr.configs.get("neus-dtu")
    st = nsr.build(cfg).cuda().train()
    grid = st.occupancy_grid_bg
    g = torch.Generator(device="cuda").manual_seed(1)
    grid.occs.copy_(torch.rand(grid.num_cells, device="cuda", generator=g) * 0.02)
    grid._binary = (torch.rand(grid._res, device="cuda", generator=g) < 0.05)
    run = FusedNeuSStep(st)
    ref = copy.deepcopy(grid)
    thre = cfg.get("grid_prune_occ_thre_bg", 0.01)
    run.refresh_bg_occupancy_async(step, occ_thre=thre)
    ob = run._occ_buf_bg
    n = int(ob["counts"][1])
    N = grid.num_cells
    # the selection drops the samples outside the unit sphere (as nerfacc does before evaluating anything) and appends the
    # rest wave by wave: the survivors' jitter is recovered from their positions
    slots = N if step < 256 else N // 4 + min(int(ref._binary.sum()), N // 4)
    cells = ob["cells"][:n].long()
    x_sel = ob["x_unit"][:3 * n].view(n, 3)
    jitter = (x_sel * ref.resolution - ref._cell_coords(cells)).clamp(0.0, 1.0)
    if step < 256:  # every cell was a candidate, slot i = cell i with jitter[i]: the survivors are exactly the inside ones
        x_all = (ref._cell_coords(torch.arange(N, device="cuda")) + ob["jitter"][:3 * N].view(N, 3)) / ref.resolution
        r_all = (x_all - 0.5).norm(dim=1)
        assert abs(n - int((r_all < 0.5).sum())) <= int(((r_all - 0.5).abs() < 1e-5).sum())
        assert int(torch.bincount(cells, minlength=N).max()) == 1
    else:
        assert 0.4 * slots < n < 0.65 * slots  # (a sphere fills 52 % of its cube)
    with torch.no_grad():
        ref._update_cells(cells, jitter, run.bg_occ_eval_fn, occ_thre=thre, ema_decay=0.95)
    once = torch.bincount(cells, minlength=N) <= 1
    # no sample outside the unit sphere survives (their cells are left alone, as in the reference); a sample within rounding
    # of the sphere's surface may be classified differently by sqrtf and torch's norm -- not compared
    x = (ref._cell_coords(cells) + jitter) / ref.resolution
    r = (x - 0.5).norm(dim=1)
    assert int((r >= 0.5 + 1e-5).sum()) == 0 and float(r.max()) > 0.49
    on_surface = torch.zeros(N, dtype=torch.bool, device="cuda")
    on_surface[cells[(r - 0.5).abs() < 1e-5]] = True
    sel = once & ~on_surface
    assert torch.allclose(grid.occs[sel], ref.occs[sel], rtol=2e-3, atol=2e-6), \
        float((grid.occs[sel] - ref.occs[sel]).abs().max())
    assert float((grid.binary != ref.binary).float().mean()) < 1e-4
    assert torch.equal(ob["bricks"], ops.grid_bricks(grid.binary.clone()))
