"""BASELINE config C5 on the MI355X: the HIP drop-in packages (through tests/refmirror, the restatement of the
reference's glue) against the fixture the REFERENCE's own models/ produced at full size (L=16, T=2^19,
ProgressiveBandHashGrid at current_level 4 / 9 / 16, finite-difference gradients + laplace, progressive eps,
fp32 VanillaMLP texture).  Reference: models/geometry.py:181-199,219-238, models/network_utils.py:40-65,
configs/neuralangelo-dtu-wmask.yaml:42-52.

Stated tolerances: segment indices bit-exact; sdf 1e-3 abs; finite-difference gradient 1e-2 abs (the taps are fp16
encodings ~eps apart: one fp16 ulp of an encoding moves a central difference by ~1e-6/eps); laplace 5 % of its
largest magnitude; colours 3e-3; masked levels EXACTLY zero in the encoding and in the table gradient; table gradient
summary 2 % of its norm; small-parameter gradients rel-L2 <= 2e-2."""
import numpy as np
import pytest
import torch

import fixture_utils as fu
from test_golden_glue import binary_from, load

pytestmark = pytest.mark.gpu
LAMBDAS = {"lambda_rgb_l1": 1.0, "lambda_mask": 0.1, "lambda_eikonal": 0.1, "lambda_sparsity": 0.01}


def _cos(a, b):
    return float(torch.nn.functional.cosine_similarity(a.flatten().double(), b.flatten().double(), dim=0))


def _model(fx):
    import nsr
    import refmirror
    cfg = nsr.configs.get("neuralangelo")
    cfg["num_samples_per_ray"] = 256
    m = refmirror.NeuSModel(cfg).cuda().train()
    sd = {k[len("param/"):]: v for k, v in fx.items() if k.startswith("param/")}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("occupancy_grid" in k or k.endswith("encoding.encoding.encoding.params") for k in missing), missing
    table = m.geometry.encoding.encoding.encoding.params
    assert table.numel() == int(fx["table_numel"])
    with torch.no_grad():
        table.copy_(fu.seeded_normal(table.numel(), int(fx["table_seed"]), std=float(fx["table_std"])).cuda())
    m.occupancy_grid._binary = binary_from(fx).cuda()
    m.background_color = fx["background"].cuda()
    m.randomized = False
    return m


@pytest.mark.parametrize("level", [4, 9, 16])
def test_neuralangelo_matches_reference_fixture(level):
    fx = load("neuralangelo_forward.npz")
    m = _model(fx)
    p = f"L{level}/"
    m.update_step(0, int(fx[p + "global_step"]))
    pg = m.geometry.encoding.encoding
    assert pg.current_level == level
    assert abs(m.geometry._finite_difference_eps - float(fx[p + "eps"])) < 1e-12
    assert abs(m.cos_anneal_ratio - float(fx[p + "cos_anneal_ratio"])) < 1e-12
    # the module-level mask: columns of levels >= current_level are exactly zero, the others match the plain encoding
    x = torch.rand(4096, 3, device="cuda")
    e = pg(x)
    assert float(e[:, 2 * level:].abs().max() if level < 16 else 0.0) == 0.0
    full = pg.encoding.level_mask_count
    pg.encoding.level_mask_count = lambda: 16
    assert torch.equal(pg.encoding(x)[:, :2 * level], e[:, :2 * level])
    pg.encoding.level_mask_count = full

    out = m(fx["rays"].cuda())
    assert torch.equal(out["ray_indices"].cpu(), fx[p + "out/ray_indices"])
    assert int(out["num_samples_full"]) == int(fx[p + "out/num_samples_full"])
    err = {}
    for k, tol in (("sdf_samples", 1e-3), ("sdf_grad_samples", 1e-2), ("comp_rgb", 3e-3), ("opacity", 3e-3),
                   ("depth", 5e-3), ("comp_rgb_full", 3e-3), ("weights", 3e-3)):
        err[k] = float((out[k].cpu() - fx[p + "out/" + k]).abs().max())
        assert err[k] <= tol, (level, k, err[k])
    lap = fx[p + "out/sdf_laplace_samples"]
    err["laplace_rel"] = float((out["sdf_laplace_samples"].cpu() - lap).abs().max() / lap.abs().max())
    assert err["laplace_rel"] <= 5e-2, (level, err)
    lam = dict(LAMBDAS, lambda_curvature=(1e-4 if level < 16 else 0.0))
    loss, terms = fu.neus_system_loss(out, fx["rgb"].cuda(), fx["fg_mask"].cuda(), lam)
    assert abs(float(loss) - float(fx[p + "loss"])) < 3e-3 * max(1.0, abs(float(fx[p + "loss"]))), (float(loss), err)
    for k in ("eikonal", "mask", "sparsity", "curvature"):
        want = float(fx[p + "term/" + k])
        assert abs(float(terms[k]) - want) <= 2e-2 * abs(want) + 1e-4, (k, float(terms[k]), want)
    loss.backward()
    params = dict(m.named_parameters())
    tkey = "geometry.encoding.encoding.encoding.params"
    g = params[tkey].grad
    off = [int(o) for o in fx["level_offsets"]]
    for l in range(level, 16):  # masked levels receive EXACTLY zero gradient
        assert float(g[off[l]:off[l + 1]].abs().max()) == 0.0, l
    fu.check_grad_summary(g, fu.unpack_summary(fx, p + "gradsum/" + tkey), rel=2e-2, name=f"table L{level}")
    for k in ("geometry.network.layers.0.weight_v", "geometry.network.layers.0.weight_g", "geometry.network.layers.2.weight_v",
              "geometry.network.layers.0.bias", "texture.network.layers.0.weight", "texture.network.layers.4.weight"):
        e = fu.rel_l2(params[k].grad.cpu(), fx[p + "grad/" + k])
        assert e < 2e-2, (level, k, e)
    gv, wv = float(params["variance.variance"].grad), float(fx[p + "grad/variance.variance"])
    assert abs(gv - wv) < 2e-2 * abs(wv) + 1e-5, (gv, wv)


def test_eval_chunked_render_matches_training_forward():
    """models/neus.py:289-296 + models/utils.py:13-50: in eval mode the rays are rendered in ``ray_chunk`` pieces, moved
    to the CPU, detached; no training-only keys; same numbers as the one-shot training forward (randomized off)"""
    fx = load("neuralangelo_forward.npz")
    m = _model(fx)
    m.update_step(0, 5005)
    rays = fx["rays"].cuda()
    ref = m(rays)
    m.eval()
    m.config["ray_chunk"] = 7  # 20 rays -> chunks of 7, 7, 6
    with torch.no_grad():
        out = m(rays)
    assert not out["comp_rgb_full"].is_cuda and not out["comp_rgb_full"].requires_grad
    assert "sdf_samples" not in out and "weights" not in out
    for k in ("comp_rgb_full", "opacity", "depth", "comp_normal"):
        assert out[k].shape == ref[k].shape
        assert torch.allclose(out[k], ref[k].detach().cpu(), atol=1e-5), k
    assert int(out["num_samples_full"].sum()) == int(ref["num_samples_full"])
    with pytest.raises(RuntimeError):  # nerfacc: every_n_step outside training raises
        m.occupancy_grid.every_n_step(step=0, occ_eval_fn=lambda x: x[:, :1])
