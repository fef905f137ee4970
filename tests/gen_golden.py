"""Mint the golden fixtures under tests/golden/ by running the REFERENCE's own glue (``/root/reference/models``,
imported unchanged) on top of the CPU oracle's tinycudann / nerfacc stand-ins.

    PYTHONDONTWRITEBYTECODE=1 python tests/gen_golden.py          (build container only: needs /root/reference)

What the fixtures pin: the reference-owned arithmetic of the hot path -- ``contract_to_unisphere``, ``trunc_exp``,
``scale_anything``, ``NeuSModel.get_alpha``, ``VolumeDensity/VolumeSDF/VolumeRadiance.forward``,
``NeRFModel.forward_`` and ``NeuSModel.forward_`` (incl. the analytic-gradient double backward) -- as executed by the
reference code itself.  tests/test_golden_glue.py checks oracle/glue_ref.py against them on CPU and
tests/test_gpu_golden.py checks the HIP path (nsr host mirror over the drop-in packages) on the MI355X.
The third-party arithmetic underneath (hash grid, MLP, marcher) is the ORACLE's here, so these fixtures do not pin
tcnn / nerfacc themselves (see oracle/__init__.py).
"""
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import refshim  # noqa: E402
import fixture_utils as fu  # noqa: E402
from oracle import nerfacc_ref, tcnn_ref  # noqa: E402

OUT = os.path.join(HERE, "golden")
SMALL_GRID = dict(n_levels=4, n_features_per_level=2, log2_hashmap_size=12, base_resolution=8, per_level_scale=2.0)


def _np(d):
    return {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in d.items()}


def _rays(n, seed):
    g = torch.Generator().manual_seed(seed)
    o = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1) * 4.0
    d = torch.nn.functional.normalize(-o + torch.randn(n, 3, generator=g) * 0.35, dim=-1)
    return torch.cat([o, d], -1)


def _valid_rays(m, rays, n_keep, key="rays_valid_full"):
    """the first n_keep of `rays` that the reference's own model renders as VALID (opacity > 0): rays are independent, so
    the selection stays valid when rendered alone -- the parity tests then compare loss and EVERY gradient
    unconditionally (the system's losses are taken over the valid rays, systems/neus.py:96-104)"""
    with torch.no_grad():
        out = m(rays)
    idx = torch.nonzero(out[key][..., 0]).flatten()[:n_keep]
    assert idx.numel() == n_keep, (idx.numel(), n_keep)
    return rays[idx].clone()


def _sphere_grid(res, radius, r_occ):
    ii = torch.stack(torch.meshgrid(*[torch.arange(res)] * 3, indexing="ij"), -1).float()
    c = (ii + 0.5) / res * 2 * radius - radius
    return c.norm(dim=-1) < r_occ


def gen_glue(models):
    from models.geometry import contract_to_unisphere
    from models.utils import get_activation, scale_anything, trunc_exp
    from nerfacc import ContractionType
    g = torch.Generator().manual_seed(0)
    x = torch.randn(256, 3, generator=g) * 3.0
    out = {"x": x, "contract_aabb": contract_to_unisphere(x.clone(), 1.5, ContractionType.AABB),
           "contract_sphere": contract_to_unisphere(x.clone(), 1.5, ContractionType.UN_BOUNDED_SPHERE),
           "scale_anything": scale_anything(x, (-1.5, 1.5), (0, 1))}
    z = (torch.randn(512, generator=g) * 6).requires_grad_(True)
    y = trunc_exp(z)
    gz = torch.randn(512, generator=g)
    y.backward(gz)
    out.update(trunc_exp_in=z, trunc_exp_out=y, trunc_exp_gin=gz, trunc_exp_grad=z.grad)
    for name in ("sigmoid", "scale2.5", "+1.5", "lin2srgb", "softplus", "none"):
        out["act_" + name] = get_activation(name)(x)
    np.savez_compressed(os.path.join(OUT, "glue_elementwise.npz"), **_np(out))


def gen_vanilla_frequency(models):
    """models/network_utils.py:14-37 -- with and without the coarse-to-fine mask, through get_encoding too"""
    from models.network_utils import VanillaFrequency, get_encoding
    g = torch.Generator().manual_seed(3)
    x = torch.randn(64, 3, generator=g)
    out = {"x": x}
    enc = VanillaFrequency(3, {"n_frequencies": 6})
    out["plain"] = enc(x)
    enc = VanillaFrequency(3, {"n_frequencies": 6, "n_masking_step": 1000})
    for step in (0, 250, 700, 5000):
        enc.update_step(0, step)
        out[f"masked_{step}"] = enc(x)
        out[f"mask_{step}"] = enc.mask.clone()
    comp = get_encoding(3, refshim.DictConfig({"otype": "VanillaFrequency", "n_frequencies": 4, "include_xyz": True}))
    out["composite"] = comp(x)
    np.savez_compressed(os.path.join(OUT, "vanilla_frequency.npz"), **_np(out))


def gen_neus_alpha(models):
    cfg = refshim.load_config("neus-blender.yaml", ["dataset.scene=lego"])
    cfg.model.geometry.xyz_encoding_config.update(SMALL_GRID)
    torch.manual_seed(0)
    m = models.make("neus", cfg.model)
    g = torch.Generator().manual_seed(1)
    n = 600
    sdf = (torch.randn(n, generator=g) * 0.05).requires_grad_(True)
    normal = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).requires_grad_(True)
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    dists = torch.rand(n, 1, generator=g) * 0.01
    out = {"sdf": sdf, "normal": normal, "dirs": dirs, "dists": dists, "inv_s": m.variance.inv_s}
    for ratio in (0.0, 0.4, 1.0):
        m.cos_anneal_ratio = ratio
        a = m.get_alpha(sdf, normal, dirs, dists)
        ga = torch.randn(n, generator=torch.Generator().manual_seed(2))
        gs, gn, gv = torch.autograd.grad(a, [sdf, normal, m.variance.variance], ga)
        out.update({f"alpha_{ratio}": a, f"g_alpha_{ratio}": ga, f"g_sdf_{ratio}": gs, f"g_normal_{ratio}": gn,
                    f"g_variance_{ratio}": gv})
    np.savez_compressed(os.path.join(OUT, "neus_alpha.npz"), **_np(out))


def gen_nerf(models):
    cfg = refshim.load_config("nerf-blender.yaml", ["dataset.scene=lego"])
    cfg.model.geometry.xyz_encoding_config.update(SMALL_GRID)
    cfg.model.num_samples_per_ray = 256
    torch.manual_seed(3)
    m = models.make("nerf", cfg.model)
    m.train()
    with torch.no_grad():
        p = m.geometry.encoding_with_network.params
        p[m.geometry.encoding_with_network.desc.n_params:].normal_(0, 0.3)
    m.occupancy_grid._binary = _sphere_grid(128, 1.5, 1.0)
    m.background_color = torch.tensor([0.2, 0.5, 0.8])
    m.randomized = False
    rays = _rays(48, 7)
    out = m(rays)
    loss = torch.nn.functional.smooth_l1_loss(out["comp_rgb"], torch.full_like(out["comp_rgb"], 0.5)) + out["depth"].mean() * 0.1
    loss.backward()
    fx = {"rays": rays, "background": m.background_color, "binary_packed": np.packbits(m.occupancy_grid._binary.numpy()),
          "render_step_size": m.render_step_size, "loss": loss}
    fx.update({"param/" + k: v for k, v in m.state_dict().items() if "occupancy" not in k})
    fx.update({"grad/" + k: v.grad for k, v in m.named_parameters() if v.grad is not None and v.numel() > 0})
    fx.update({"out/" + k: v for k, v in out.items()})
    # field-level outputs on fixed points (VolumeDensity.forward / VolumeRadiance.forward)
    pts = (torch.rand(300, 3, generator=torch.Generator().manual_seed(9)) - 0.5) * 2.4
    dens, feat = m.geometry(pts)
    dirs = torch.nn.functional.normalize(torch.randn(300, 3, generator=torch.Generator().manual_seed(10)), dim=-1)
    fx.update({"field/points": pts, "field/dirs": dirs, "field/density": dens, "field/feature": feat,
               "field/rgb": m.texture(feat, dirs)})
    np.savez_compressed(os.path.join(OUT, "nerf_forward.npz"), **_np(fx))


def gen_neus(models):
    cfg = refshim.load_config("neus-blender.yaml", ["dataset.scene=lego"])
    cfg.model.geometry.xyz_encoding_config.update(SMALL_GRID)
    cfg.model.num_samples_per_ray = 256
    torch.manual_seed(4)
    m = models.make("neus", cfg.model)
    m.train()
    m.update_step(0, 5000)  # cos_anneal_ratio = 0.25; occupancy refresh is skipped (5000 % 16 != 0)
    with torch.no_grad():
        m.geometry.encoding.encoding.params.normal_(0, 0.05)
        # sphere_init zeroes the first layer's weights on the encoding columns (network_utils.py:121-123); give them
        # mass so the fixture exercises the hash-grid gradients (first AND second order), as a trained model does
        m.geometry.network.layers[0].weight_v[:, 3:].normal_(0, 0.05)
    m.occupancy_grid._binary = _sphere_grid(128, 1.5, 0.8)
    m.background_color = torch.tensor([1.0, 1.0, 1.0])
    m.randomized = False
    rays = _valid_rays(m, _rays(48, 11), 24)
    out = m(rays)
    assert bool(out["rays_valid_full"].all())
    eik = ((torch.linalg.norm(out["sdf_grad_samples"], ord=2, dim=-1) - 1.0) ** 2).mean()  # systems/neus.py:106
    loss = torch.nn.functional.mse_loss(out["comp_rgb_full"], torch.full_like(out["comp_rgb_full"], 0.4)) * 10 + eik * 0.1
    loss.backward()
    fx = {"rays": rays, "background": m.background_color, "binary_packed": np.packbits(m.occupancy_grid._binary.numpy()),
          "cos_anneal_ratio": m.cos_anneal_ratio, "loss": loss, "loss_eikonal": eik}
    fx.update({"param/" + k: v for k, v in m.state_dict().items() if "occupancy" not in k})
    fx.update({"grad/" + k: v.grad for k, v in m.named_parameters() if v.grad is not None and v.numel() > 0})
    fx.update({"out/" + k: v for k, v in out.items()})
    np.savez_compressed(os.path.join(OUT, "neus_forward.npz"), **_np(fx))


def gen_neus_bg(models):
    """configs/neus-dtu.yaml (C4): NeuS foreground + learned NeRF++ background (models/neus.py:141-203, 259-287)"""
    cfg = refshim.load_config("neus-dtu.yaml", ["dataset.root_dir=unused"])
    cfg.model.geometry.xyz_encoding_config.update(SMALL_GRID)
    cfg.model.geometry_bg.xyz_encoding_config.update(SMALL_GRID)
    cfg.model.num_samples_per_ray = 256
    torch.manual_seed(5)
    m = models.make("neus", cfg.model)
    m.train()
    m.update_step(0, 5000)  # no occupancy refresh (5000 % 16 != 0)
    with torch.no_grad():
        m.geometry.encoding.encoding.params.normal_(0, 0.05)
        m.geometry.network.layers[0].weight_v[:, 3:].normal_(0, 0.05)
        m.geometry_bg.encoding_with_network.encoding.encoding.params.normal_(0, 0.3)
    m.occupancy_grid._binary = _sphere_grid(128, 1.0, 0.6)
    ii = torch.stack(torch.meshgrid(*[torch.arange(256)] * 3, indexing="ij"), -1)
    m.occupancy_grid_bg._binary = ((ii.sum(-1) % 3) != 0)  # a deterministic 2/3-full pattern of the contracted space
    m.background_color = torch.tensor([1.0, 1.0, 1.0])
    m.randomized = False
    rays = _rays(32, 12)
    rays[:, :3] *= 0.6  # cameras at radius 2.4 around the radius-1 foreground box
    rays = _valid_rays(m, rays, 16)
    out = m(rays)
    assert bool(out["rays_valid_full"].all())
    eik = ((torch.linalg.norm(out["sdf_grad_samples"], ord=2, dim=-1) - 1.0) ** 2).mean()
    loss = torch.nn.functional.mse_loss(out["comp_rgb_full"], torch.full_like(out["comp_rgb_full"], 0.4)) * 10 + eik * 0.1
    loss.backward()
    fx = {"rays": rays, "background": m.background_color, "binary_packed": np.packbits(m.occupancy_grid._binary.numpy()),
          "binary_bg_packed": np.packbits(m.occupancy_grid_bg._binary.numpy()),
          "cos_anneal_ratio": m.cos_anneal_ratio, "loss": loss, "loss_eikonal": eik}
    fx.update({"param/" + k: v for k, v in m.state_dict().items() if "occupancy" not in k})
    fx.update({"grad/" + k: v.grad for k, v in m.named_parameters() if v.grad is not None and v.numel() > 0})
    fx.update({"out/" + k: v for k, v in out.items()})
    np.savez_compressed(os.path.join(OUT, "neus_bg_forward.npz"), **_np(fx))


NEURALANGELO_STEPS = {4: 5, 9: 5005, 16: 12005}  # current_level -> a global step that is not an occupancy refresh
NEURALANGELO_LAMBDAS = {"lambda_rgb_l1": 1.0, "lambda_mask": 0.1, "lambda_eikonal": 0.1, "lambda_sparsity": 0.01}


def gen_neuralangelo(models):
    """configs/neuralangelo-dtu-wmask.yaml (C5) at FULL size (L=16, T=2^19, include_xyz): ProgressiveBandHashGrid at
    current_level 4 / 9 / 16, finite-difference gradients + laplace with the progressive eps
    (models/geometry.py:181-199,219-238, models/network_utils.py:40-65), fp32 VanillaMLP texture, and the loss terms of
    systems/neus.py.  The 14 M-entry table is re-generated from a seed on the test side; its gradient is pinned by a
    summary (tests/fixture_utils.py)."""
    cfg = refshim.load_config("neuralangelo-dtu-wmask.yaml", ["dataset.root_dir=unused"])
    cfg.model.num_samples_per_ray = 256
    torch.manual_seed(6)
    m = models.make("neus", cfg.model)
    m.train()
    table = m.geometry.encoding.encoding.encoding.params
    desc = m.geometry.encoding.encoding.encoding.desc
    with torch.no_grad():
        table.copy_(fu.seeded_normal(table.numel(), 606, std=0.05))
        m.geometry.network.layers[0].weight_v[:, 3:].copy_(fu.seeded_normal(64 * 32, 607, std=0.05).view(64, 32))
    m.occupancy_grid._binary = _sphere_grid(128, 1.0, 0.6)
    m.background_color = torch.tensor([1.0, 1.0, 1.0])
    m.randomized = False
    rays = _rays(20, 13)
    rays[:, :3] *= 0.6
    g = torch.Generator().manual_seed(14)
    rgb = torch.rand(20, 3, generator=g)
    fg_mask = (torch.rand(20, generator=g) > 0.3).float()
    offsets = [int(o) * desc.F for o in desc.offset]
    fx = {"rays": rays, "rgb": rgb, "fg_mask": fg_mask, "background": m.background_color,
          "binary_packed": np.packbits(m.occupancy_grid._binary.numpy()), "table_seed": 606, "table_std": 0.05,
          "table_numel": table.numel(), "level_offsets": np.asarray(offsets)}
    fx.update({"param/" + k: v for k, v in m.state_dict().items()
               if "occupancy" not in k and v.numel() < 100000})
    for level, step in NEURALANGELO_STEPS.items():
        m.zero_grad(set_to_none=True)
        m.update_step(0, step)
        assert m.geometry.encoding.encoding.current_level == level
        out = m(rays)
        lam = dict(NEURALANGELO_LAMBDAS, lambda_curvature=(1e-4 if level < 16 else 0.0))
        loss, terms = fu.neus_system_loss(out, rgb, fg_mask, lam)
        loss.backward()
        p = f"L{level}/"
        fx.update({p + "global_step": step, p + "eps": m.geometry._finite_difference_eps,
                   p + "cos_anneal_ratio": m.cos_anneal_ratio, p + "loss": loss,
                   p + "mask": m.geometry.encoding.encoding.mask.clone()})
        fx.update({p + "term/" + k: v for k, v in terms.items()})
        fx.update({p + "out/" + k: v for k, v in out.items()})
        for k, v in m.named_parameters():
            if v.grad is None or v.numel() == 0:
                continue
            if v.numel() < 100000:
                fx[p + "grad/" + k] = v.grad.clone()
            else:
                fx.update(fu.pack_summary(p + "gradsum/" + k, fu.grad_summary(v.grad, offsets)))
    np.savez_compressed(os.path.join(OUT, "neuralangelo_forward.npz"), **_np(fx))


from gen_golden_constants import FULL_LAMBDAS  # noqa: E402


def _gen_full(models, yaml, overrides, out_name, seed, with_bg):
    """configs/neus-blender.yaml (C3) / configs/neus-dtu.yaml (C4) at FULL size (L=16, T=2^19): analytic normals with the
    double backward, the system's loss terms on the valid rays (systems/neus.py:96-130 via fixture_utils.neus_system_loss);
    tables re-generated from seeds on the test side, their 14 M-entry gradients pinned by summaries (the scheme of
    gen_neuralangelo)"""
    cfg = refshim.load_config(yaml, overrides)
    cfg.model.num_samples_per_ray = 256
    torch.manual_seed(seed)
    m = models.make("neus", cfg.model)
    m.train()
    m.update_step(0, 5000)  # cos_anneal_ratio = 0.25; no occupancy refresh (5000 % 16 != 0)
    enc = m.geometry.encoding.encoding
    seeds = {"geometry.encoding.encoding.params": (seed * 100 + 1, 0.05)}
    with torch.no_grad():
        enc.params.copy_(fu.seeded_normal(enc.params.numel(), seed * 100 + 1, std=0.05))
        m.geometry.network.layers[0].weight_v[:, 3:].copy_(fu.seeded_normal(64 * 32, seed * 100 + 2, std=0.05).view(64, 32))
        if with_bg:
            eb = m.geometry_bg.encoding_with_network.encoding.encoding
            eb.params.copy_(fu.seeded_normal(eb.params.numel(), seed * 100 + 3, std=0.3))
            seeds["geometry_bg.encoding_with_network.encoding.encoding.params"] = (seed * 100 + 3, 0.3)
    r = float(cfg.model.radius)
    m.occupancy_grid._binary = _sphere_grid(128, r, 0.6 * r / 1.0 if with_bg else 0.8)
    if with_bg:
        ii = torch.stack(torch.meshgrid(*[torch.arange(256)] * 3, indexing="ij"), -1)
        m.occupancy_grid_bg._binary = ((ii.sum(-1) % 3) != 0)
    m.background_color = torch.tensor([1.0, 1.0, 1.0])
    m.randomized = False
    rays = _rays(40, seed + 20)
    if with_bg:
        rays[:, :3] *= 0.6
    rays = _valid_rays(m, rays, 20)
    g = torch.Generator().manual_seed(seed + 30)
    rgb = torch.rand(20, 3, generator=g)
    fg_mask = (torch.rand(20, generator=g) > 0.3).float()
    out = m(rays)
    assert bool(out["rays_valid_full"].all())
    loss, terms = fu.neus_system_loss(out, rgb, fg_mask, FULL_LAMBDAS)
    loss.backward()
    desc = enc.desc
    offsets = [int(o) * desc.F for o in desc.offset]
    fx = {"rays": rays, "rgb": rgb, "fg_mask": fg_mask, "background": m.background_color,
          "binary_packed": np.packbits(m.occupancy_grid._binary.numpy()), "level_offsets": np.asarray(offsets),
          "cos_anneal_ratio": m.cos_anneal_ratio, "loss": loss}
    if with_bg:
        fx["binary_bg_packed"] = np.packbits(m.occupancy_grid_bg._binary.numpy())
    for k, (sd, std) in seeds.items():
        fx.update({"seed/" + k: sd, "std/" + k: std, "numel/" + k: dict(m.named_parameters())[k].numel()})
    fx.update({"param/" + k: v for k, v in m.state_dict().items() if "occupancy" not in k and v.numel() < 100000})
    fx.update({"term/" + k: v for k, v in terms.items()})
    fx.update({"out/" + k: v for k, v in out.items()})
    for k, v in m.named_parameters():
        if v.grad is None or v.numel() == 0:
            continue
        if v.numel() < 100000:
            fx["grad/" + k] = v.grad.clone()
        else:
            fx.update(fu.pack_summary("gradsum/" + k, fu.grad_summary(v.grad, offsets)))
    np.savez_compressed(os.path.join(OUT, out_name), **_np(fx))


def gen_neus_full(models):
    _gen_full(models, "neus-blender.yaml", ["dataset.scene=lego"], "neus_full_forward.npz", 8, False)


def gen_neus_bg_full(models):
    _gen_full(models, "neus-dtu.yaml", ["dataset.root_dir=unused"], "neus_bg_full_forward.npz", 9, True)


def gen_boundary_traces():
    """tests/trace_tools.py: the reference's models from the REAL YAMLs (full-size C2 / C3) on recording wrappers of the
    oracle packages -> tests/golden/trace_{nerf,neus}.npz"""
    import trace_tools
    for name, yaml_name, cli, seed in (("nerf", "nerf-blender.yaml", ["dataset.scene=lego"], 21),
                                       ("neus", "neus-blender.yaml", ["dataset.scene=lego"], 22)):
        rec = trace_tools.Recorder()
        models = refshim.install(rec.wrap_tcnn(tcnn_ref), rec.wrap_nerfacc(nerfacc_ref))
        try:
            cfg = refshim.load_config(yaml_name, cli)
            torch.manual_seed(seed)
            m = models.make(name, cfg.model)
            m.train()

            def plan(info, mod):
                if info["cls"] == "NetworkWithInputEncoding":
                    n_net = mod.desc.n_params
                    return [(n_net, seed * 100 + 1, 0.2), (mod.params.numel() - n_net, seed * 100 + 2, 0.3)]
                if info["cls"] == "Network":
                    return [(mod.params.numel(), seed * 100 + 3, 0.2)]
                return [(mod.params.numel(), seed * 100 + 4, 0.05)]
            rec.seed_parameters(m, plan)
            if name == "neus":
                with torch.no_grad():
                    m.geometry.network.layers[0].weight_v[:, 3:].copy_(fu.seeded_normal(64 * 32, seed * 100 + 5, 0.05).view(64, 32))
            m.update_step(0, 5)  # not an occupancy-refresh step
            m.occupancy_grid._binary = _sphere_grid(128, 1.5, 0.5)
            m.background_color = torch.tensor([0.1, 0.6, 0.9])
            m.randomized = False
            rays = _rays(12, seed)
            out = m(rays)
            if name == "nerf":
                valid = out["rays_valid"][..., 0]
                loss = torch.nn.functional.smooth_l1_loss(out["comp_rgb"][valid], torch.full_like(out["comp_rgb"], 0.5)[valid])
            else:
                rgb = torch.full_like(out["comp_rgb_full"], 0.5)
                loss, _ = fu.neus_system_loss(out, rgb, torch.ones(12), {"lambda_rgb_l1": 1.0, "lambda_eikonal": 0.1,
                                                                         "lambda_mask": 0.1})
            loss.backward()
            fx = rec.finish(m, extra={"model": name, "yaml": yaml_name, "loss": float(loss),
                                      "num_samples": int(out["num_samples"].sum())})
            np.savez_compressed(os.path.join(OUT, f"trace_{name}.npz"), **fx)
            print(name, "calls:", [(c["kind"], c.get("fn", c.get("mod"))) for c in rec.calls])
        finally:
            refshim.uninstall()


def main():
    os.makedirs(OUT, exist_ok=True)
    models = refshim.install(tcnn_ref, nerfacc_ref)
    gen_glue(models)
    gen_vanilla_frequency(models)
    gen_neus_alpha(models)
    gen_nerf(models)
    gen_neus(models)
    gen_neus_bg(models)
    gen_neuralangelo(models)
    gen_neus_full(models)
    gen_neus_bg_full(models)
    refshim.uninstall()
    gen_boundary_traces()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    if len(sys.argv) > 1:  # only the named generators, e.g.  python tests/gen_golden.py gen_neus gen_neus_full
        os.makedirs(OUT, exist_ok=True)
        models_ = refshim.install(tcnn_ref, nerfacc_ref)
        for name in sys.argv[1:]:
            globals()[name](models_)
        refshim.uninstall()
    else:
        main()
