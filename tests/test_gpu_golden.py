"""The HIP path (host mirror over the drop-in tinycudann / nerfacc packages) against the golden fixtures that the
REFERENCE's glue produced (tests/gen_golden.py).  Tolerances: segment indices and sample positions bit exact;
rendered colours 2e-3 (fp16 fields), SDF 1e-3, gradients cosine >= 0.999."""
import copy

import numpy as np

import pytest
import torch

import fixture_utils as fu

from test_golden_glue import SMALL_GRID, binary_from, load

pytestmark = pytest.mark.gpu


def _cos(a, b):
    return float(torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0))


def test_nerf_model_matches_reference_fixture():
    import nsr
    import refmirror
    fx = load("nerf_forward.npz")
    cfg = nsr.configs.get("nerf-blender")
    cfg["geometry"]["xyz_encoding_config"].update({k: v for k, v in SMALL_GRID.items() if k != "otype"})
    cfg["num_samples_per_ray"] = 256
    m = refmirror.NeRFModel(cfg).cuda().train()
    sd = {k[len("param/"):]: v for k, v in fx.items() if k.startswith("param/")}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("occupancy_grid" in k for k in missing), (missing, unexpected)
    m.occupancy_grid._binary = binary_from(fx).cuda()
    m.background_color = fx["background"].cuda()
    m.randomized = False
    assert abs(m.render_step_size - float(fx["render_step_size"])) < 1e-12
    out = m(fx["rays"].cuda())
    assert torch.equal(out["ray_indices"].cpu(), fx["out/ray_indices"])
    assert torch.equal(out["points"].cpu(), fx["out/points"]) and torch.equal(out["intervals"].cpu(), fx["out/intervals"])
    assert out["num_samples"].dtype == torch.int32 and int(out["num_samples"]) == int(fx["out/num_samples"])
    for k in ("comp_rgb", "opacity", "depth"):
        assert out[k].shape == fx["out/" + k].shape
        assert torch.allclose(out[k].cpu(), fx["out/" + k], atol=2e-3), (k, (out[k].cpu() - fx["out/" + k]).abs().max())
    assert torch.equal(out["rays_valid"].cpu(), fx["out/rays_valid"])
    loss = torch.nn.functional.smooth_l1_loss(out["comp_rgb"], torch.full_like(out["comp_rgb"], 0.5)) + out["depth"].mean() * 0.1
    loss.backward()
    assert abs(float(loss) - float(fx["loss"])) < 1e-3
    for k in ("geometry.encoding_with_network.params", "texture.network.params"):
        g = dict(m.named_parameters())[k].grad.cpu()
        fu.assert_grad(g, fx["grad/" + k], k)
    dens, feat = m.geometry(fx["field/points"].cuda())
    assert torch.allclose(dens.cpu(), fx["field/density"], rtol=5e-3, atol=1e-4)
    assert torch.allclose(feat.cpu(), fx["field/feature"], rtol=5e-3, atol=2e-3)
    rgb = m.texture(feat, fx["field/dirs"].cuda())
    assert torch.allclose(rgb.cpu(), fx["field/rgb"], atol=3e-3)


def test_neus_model_matches_reference_fixture():
    import nsr
    import refmirror
    fx = load("neus_forward.npz")
    cfg = nsr.configs.get("neus-blender")
    cfg["geometry"]["xyz_encoding_config"].update({k: v for k, v in SMALL_GRID.items() if k != "otype"})
    cfg["num_samples_per_ray"] = 256
    m = refmirror.NeuSModel(cfg).cuda().train()
    sd = {k[len("param/"):]: v for k, v in fx.items() if k.startswith("param/")}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("occupancy_grid" in k for k in missing), (missing, unexpected)
    m.update_step(0, 5000)
    assert abs(m.cos_anneal_ratio - float(fx["cos_anneal_ratio"])) < 1e-9
    m.occupancy_grid._binary = binary_from(fx).cuda()
    m.background_color = fx["background"].cuda()
    m.randomized = False
    out = m(fx["rays"].cuda())
    assert torch.equal(out["ray_indices"].cpu(), fx["out/ray_indices"])
    assert torch.allclose(out["sdf_samples"].cpu(), fx["out/sdf_samples"], atol=1e-3)
    assert torch.allclose(out["sdf_grad_samples"].cpu(), fx["out/sdf_grad_samples"], atol=2e-2)
    for k in ("comp_rgb", "opacity", "depth", "comp_rgb_full"):
        assert torch.allclose(out[k].cpu(), fx["out/" + k], atol=3e-3), (k, (out[k].cpu() - fx["out/" + k]).abs().max())
    assert torch.allclose(out["inv_s"].cpu(), torch.exp(fx["param/variance.variance"] * 10.0))
    eik = ((torch.linalg.norm(out["sdf_grad_samples"], ord=2, dim=-1) - 1.0) ** 2).mean()
    loss = torch.nn.functional.mse_loss(out["comp_rgb_full"], torch.full_like(out["comp_rgb_full"], 0.4)) * 10 + eik * 0.1
    loss.backward()
    assert abs(float(eik) - float(fx["loss_eikonal"])) < 2e-3 * max(1.0, float(fx["loss_eikonal"]))
    params = dict(m.named_parameters())
    for k in ("geometry.encoding.encoding.params", "texture.network.params", "geometry.network.layers.0.weight_v",
              "geometry.network.layers.2.weight_v"):
        fu.assert_grad(params[k].grad, fx["grad/" + k], k)
    assert abs(float(params["variance.variance"].grad) - float(fx["grad/variance.variance"])) < \
        2e-2 * abs(float(fx["grad/variance.variance"])) + 1e-5


def test_neus_background_model_matches_reference_fixture():
    """C4 shapes (configs/neus-dtu.yaml): NeuS foreground + learned NeRF++ background -- forward_bg_ (models/neus.py:141-203:
    cone-angle marching through the UN_BOUNDED_SPHERE grid from the foreground box's exit, VanillaMLP density/colour heads)
    and the full composite of models/neus.py:259-287, against the reference's own run of it"""
    import nsr
    import refmirror
    fx = load("neus_bg_forward.npz")
    cfg = nsr.configs.get("neus-dtu")
    for key in ("geometry", "geometry_bg"):
        cfg[key]["xyz_encoding_config"].update({k: v for k, v in SMALL_GRID.items() if k != "otype"})
    cfg["num_samples_per_ray"] = 256
    m = refmirror.NeuSModel(cfg).cuda().train()
    sd = {k[len("param/"):]: v for k, v in fx.items() if k.startswith("param/")}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("occupancy_grid" in k for k in missing), (missing, unexpected)
    m.update_step(0, 5000)
    m.occupancy_grid._binary = binary_from(fx).cuda()
    m.occupancy_grid_bg._binary = torch.from_numpy(np.unpackbits(fx["binary_bg_packed"].numpy())[:256 ** 3]
                                                   .reshape(256, 256, 256).astype(bool)).cuda()
    m.background_color = fx["background"].cuda()
    m.randomized = False
    out = m(fx["rays"].cuda())
    assert torch.equal(out["ray_indices"].cpu(), fx["out/ray_indices"])
    assert torch.equal(out["ray_indices_bg"].cpu(), fx["out/ray_indices_bg"])          # same marcher decisions
    assert torch.allclose(out["points_bg"].cpu(), fx["out/points_bg"], rtol=1e-6, atol=1e-6)
    assert torch.allclose(out["intervals_bg"].cpu(), fx["out/intervals_bg"], rtol=1e-5, atol=1e-7)
    for k in ("comp_rgb_bg", "opacity_bg", "depth_bg", "comp_rgb", "opacity", "comp_rgb_full"):
        tol = 2e-2 if k == "depth_bg" else 3e-3
        assert torch.allclose(out[k].cpu(), fx["out/" + k], atol=tol, rtol=1e-2), \
            (k, (out[k].cpu() - fx["out/" + k]).abs().max())
    assert torch.equal(out["rays_valid_full"].cpu(), fx["out/rays_valid_full"])
    assert int(out["num_samples_full"]) == int(fx["out/num_samples_full"])
    eik = ((torch.linalg.norm(out["sdf_grad_samples"], ord=2, dim=-1) - 1.0) ** 2).mean()
    loss = torch.nn.functional.mse_loss(out["comp_rgb_full"], torch.full_like(out["comp_rgb_full"], 0.4)) * 10 + eik * 0.1
    loss.backward()
    assert abs(float(loss) - float(fx["loss"])) < 2e-3 * max(1.0, float(fx["loss"]))
    params = dict(m.named_parameters())
    for k in ("geometry_bg.encoding_with_network.encoding.encoding.params",
              "geometry_bg.encoding_with_network.network.layers.0.weight", "texture_bg.network.layers.4.weight",
              "geometry.encoding.encoding.params", "texture.network.layers.0.weight"):
        fu.assert_grad(params[k].grad, fx["grad/" + k], k)
