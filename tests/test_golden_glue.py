"""The committed golden fixtures (outputs of the REFERENCE's own glue, tests/gen_golden.py) against
(a) the oracle's restatement of that glue (oracle/glue_ref.py) and (b) the pure-torch pieces of the host mirror.
CPU only.  This is what pins the reference-owned arithmetic of the hot path."""
import os

import numpy as np
import pytest
import torch

from oracle import glue_ref
from oracle import nerfacc_ref as N
from oracle import tcnn_ref as T

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SMALL_GRID = dict(otype="HashGrid", n_levels=4, n_features_per_level=2, log2_hashmap_size=12, base_resolution=8,
                  per_level_scale=2.0)


def load(name):
    z = np.load(os.path.join(GOLD, name))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def binary_from(fx, res=128):
    return torch.from_numpy(np.unpackbits(fx["binary_packed"].numpy())[:res ** 3].astype(bool)).view(res, res, res)


def test_elementwise_glue():
    fx = load("glue_elementwise.npz")
    x = fx["x"]
    assert torch.equal(glue_ref.contract_to_unisphere(x, 1.5, N.ContractionType.AABB), fx["contract_aabb"])
    assert torch.allclose(glue_ref.contract_to_unisphere(x, 1.5, N.ContractionType.UN_BOUNDED_SPHERE),
                          fx["contract_sphere"], atol=1e-7)
    assert torch.equal(glue_ref.scale_anything(x, (-1.5, 1.5), (0, 1)), fx["scale_anything"])
    z = fx["trunc_exp_in"].clone().requires_grad_(True)
    y = glue_ref.trunc_exp(z)
    y.backward(fx["trunc_exp_gin"])
    assert torch.equal(y.detach(), fx["trunc_exp_out"]) and torch.equal(z.grad, fx["trunc_exp_grad"])


def test_host_mirror_activations_and_contraction_match_reference():
    from nerfacc import ContractionType
    from refmirror.fields import contract_to_unisphere, get_activation, trunc_exp
    fx = load("glue_elementwise.npz")
    x = fx["x"]
    for name in ("sigmoid", "scale2.5", "+1.5", "softplus", "none"):
        assert torch.allclose(get_activation(name)(x), fx["act_" + name], atol=1e-7), name
    z = fx["trunc_exp_in"].clone().requires_grad_(True)
    y = trunc_exp(z)
    y.backward(fx["trunc_exp_gin"])
    assert torch.equal(y.detach(), fx["trunc_exp_out"]) and torch.equal(z.grad, fx["trunc_exp_grad"])
    xr = x.clone().requires_grad_(True)  # differentiable (torch) branch of the mirror
    assert torch.allclose(contract_to_unisphere(xr, 1.5, ContractionType.AABB), fx["contract_aabb"], atol=1e-7)
    assert torch.allclose(contract_to_unisphere(xr, 1.5, ContractionType.UN_BOUNDED_SPHERE), fx["contract_sphere"],
                          atol=1e-7)


def test_neus_alpha():
    fx = load("neus_alpha.npz")
    for ratio in (0.0, 0.4, 1.0):
        sdf = fx["sdf"].clone().requires_grad_(True)
        normal = fx["normal"].clone().requires_grad_(True)
        variance = torch.tensor(0.3, requires_grad=True)
        a = glue_ref.neus_alpha(sdf, normal, fx["dirs"], fx["dists"], torch.exp(variance * 10.0), ratio)
        assert torch.allclose(a, fx[f"alpha_{ratio}"], atol=1e-7)
        gs, gn, gv = torch.autograd.grad(a, [sdf, normal, variance], fx[f"g_alpha_{ratio}"])
        assert torch.allclose(gs, fx[f"g_sdf_{ratio}"], rtol=1e-5, atol=1e-6)
        assert torch.allclose(gn, fx[f"g_normal_{ratio}"], rtol=1e-5, atol=1e-7)
        assert torch.allclose(gv, fx[f"g_variance_{ratio}"], rtol=1e-4, atol=1e-6)


def _oracle_nerf(fx):
    ewn = T.NetworkWithInputEncoding(3, 16, SMALL_GRID, dict(otype="FullyFusedMLP", activation="ReLU",
                                                             output_activation="none", n_neurons=64, n_hidden_layers=1))
    sh = T.Encoding(3, dict(otype="SphericalHarmonics", degree=4))
    net = T.Network(32, 3, dict(otype="FullyFusedMLP", activation="ReLU", output_activation="Sigmoid", n_neurons=64,
                                n_hidden_layers=2))
    with torch.no_grad():
        ewn.params.copy_(fx["param/geometry.encoding_with_network.params"])
        net.params.copy_(fx["param/texture.network.params"])
    return ewn, sh, net


def test_nerf_forward_and_fields():
    fx = load("nerf_forward.npz")
    ewn, sh, net = _oracle_nerf(fx)
    grid = N.OccupancyGrid(fx["param/scene_aabb"], 128)
    grid._binary = binary_from(fx)
    out = glue_ref.nerf_forward(fx["rays"], ewn, sh, net, grid, fx["param/scene_aabb"], 1.5,
                                float(fx["render_step_size"]), fx["background"])
    assert torch.equal(out["ray_indices"], fx["out/ray_indices"])          # bit exact segment indices
    assert torch.equal(out["points"], fx["out/points"]) and torch.equal(out["intervals"], fx["out/intervals"])
    for k in ("comp_rgb", "opacity", "depth", "weights"):
        assert torch.allclose(out[k], fx["out/" + k], atol=2e-6), k
    assert torch.equal(out["rays_valid"], fx["out/rays_valid"]) and int(out["num_samples"]) == int(fx["out/num_samples"])
    loss = torch.nn.functional.smooth_l1_loss(out["comp_rgb"], torch.full_like(out["comp_rgb"], 0.5)) + out["depth"].mean() * 0.1
    loss.backward()
    assert abs(float(loss) - float(fx["loss"])) < 1e-6
    assert torch.allclose(ewn.params.grad, fx["grad/geometry.encoding_with_network.params"], rtol=1e-3, atol=1e-7)
    assert torch.allclose(net.params.grad, fx["grad/texture.network.params"], rtol=1e-3, atol=1e-7)
    dens, feat = glue_ref.volume_density(fx["field/points"], ewn, 1.5, N.ContractionType.AABB)
    assert torch.allclose(dens, fx["field/density"], rtol=1e-6) and torch.equal(feat, fx["field/feature"])
    assert torch.allclose(glue_ref.volume_radiance(feat, fx["field/dirs"], sh, net), fx["field/rgb"], atol=1e-6)


def test_neus_forward_with_eikonal_double_backward():
    fx = load("neus_forward.npz")
    enc = T.Encoding(3, SMALL_GRID)
    sdf_mlp = torch.nn.Sequential(torch.nn.utils.weight_norm(torch.nn.Linear(11, 64)), torch.nn.Softplus(beta=100),
                                  torch.nn.utils.weight_norm(torch.nn.Linear(64, 13)))
    sh = T.Encoding(3, dict(otype="SphericalHarmonics", degree=4))
    net = T.Network(32, 3, dict(otype="FullyFusedMLP", activation="ReLU", output_activation="none", n_neurons=64,
                                n_hidden_layers=2))
    variance = torch.tensor(float(fx["param/variance.variance"]), requires_grad=True)
    with torch.no_grad():
        enc.params.copy_(fx["param/geometry.encoding.encoding.params"])
        net.params.copy_(fx["param/texture.network.params"])
        sdf_mlp.load_state_dict({k[len("param/geometry.network.layers."):]: v for k, v in fx.items()
                                 if k.startswith("param/geometry.network.layers.")})
    grid = N.OccupancyGrid(fx["param/scene_aabb"], 128)
    grid._binary = binary_from(fx)
    step = 1.732 * 2 * 1.5 / 256
    out = glue_ref.neus_forward(fx["rays"], enc, sdf_mlp, sh, net, torch.exp(variance * 10.0), grid,
                                fx["param/scene_aabb"], 1.5, step, float(fx["cos_anneal_ratio"]), fx["background"])
    assert torch.equal(out["ray_indices"], fx["out/ray_indices"])
    for k in ("sdf_samples", "sdf_grad_samples", "comp_rgb", "comp_normal", "opacity", "depth", "weights",
              "comp_rgb_full"):
        assert torch.allclose(out[k], fx["out/" + k], rtol=1e-5, atol=3e-6), k
    eik = ((torch.linalg.norm(out["sdf_grad_samples"], ord=2, dim=-1) - 1.0) ** 2).mean()
    loss = torch.nn.functional.mse_loss(out["comp_rgb_full"], torch.full_like(out["comp_rgb_full"], 0.4)) * 10 + eik * 0.1
    loss.backward()
    assert abs(float(loss) - float(fx["loss"])) < 1e-5 and abs(float(eik) - float(fx["loss_eikonal"])) < 1e-6
    assert torch.allclose(enc.params.grad, fx["grad/geometry.encoding.encoding.params"], rtol=2e-3, atol=1e-6)
    assert torch.allclose(net.params.grad, fx["grad/texture.network.params"], rtol=2e-3, atol=1e-6)
    assert torch.allclose(variance.grad, fx["grad/variance.variance"], rtol=1e-3, atol=1e-6)
    assert torch.allclose(sdf_mlp[0].weight_v.grad, fx["grad/geometry.network.layers.0.weight_v"], rtol=2e-3, atol=1e-6)


@pytest.mark.skipif(not os.path.isdir("/root/reference/models"), reason="reference tree only exists in the build container")
def test_reference_models_import_unchanged_on_the_oracle_backends():
    import refshim
    try:
        _check_reference_import(refshim)
    finally:
        refshim.uninstall()


def _check_reference_import(refshim):
    models = refshim.install(T, N)
    assert set(models.models) >= {"nerf", "neus", "volume-density", "volume-sdf", "volume-radiance", "volume-color"}
    cfg = refshim.load_config("nerf-blender.yaml", ["dataset.scene=lego"])
    assert cfg.model.geometry.xyz_encoding_config.n_levels == 16 and cfg.model.texture.input_feature_dim == 16
    cfg.model.geometry.xyz_encoding_config.update(dict(n_levels=4, log2_hashmap_size=12, base_resolution=8))
    m = models.make("nerf", cfg.model)
    assert [n for n, _ in m.named_parameters()] == ["geometry.encoding_with_network.params",
                                                    "texture.encoding.encoding.params", "texture.network.params"]


def test_vanilla_frequency_mirror_matches_reference():
    """refmirror.fields.VanillaFrequency == the reference module, masks included, via get_encoding too"""
    from refmirror.fields import VanillaFrequency, get_encoding
    fx = load("vanilla_frequency.npz")
    x = fx["x"]
    assert torch.equal(VanillaFrequency(3, {"n_frequencies": 6})(x), fx["plain"])
    enc = VanillaFrequency(3, {"n_frequencies": 6, "n_masking_step": 1000})
    assert enc.n_output_dims == 36 and enc.n_input_dims == 3
    for step in (0, 250, 700, 5000):
        enc.update_step(0, step)
        assert torch.equal(enc.mask, fx[f"mask_{step}"]), step
        assert torch.equal(enc(x), fx[f"masked_{step}"]), step
    comp = get_encoding(3, {"otype": "VanillaFrequency", "n_frequencies": 4, "include_xyz": True})
    assert comp.n_output_dims == 27 and torch.equal(comp(x), fx["composite"])
    xg = x.clone().requires_grad_(True)  # stays differentiable to second order (eikonal term of the NeuS systems)
    g1, = torch.autograd.grad(enc(xg).sum(), xg, create_graph=True)
    g2, = torch.autograd.grad(g1.sum(), xg)
    assert g2.shape == x.shape and bool(torch.isfinite(g2).all())
