"""BASELINE config C5 (configs/neuralangelo-dtu-wmask.yaml: progressive hash levels + finite-difference eikonal) at
FULL size -- the fixture tests/golden/neuralangelo_forward.npz was produced by the REFERENCE's own models/ on the oracle
backends (tests/gen_golden.py:gen_neuralangelo).  CPU: oracle/glue_ref.py's restatement against it.
(GPU twin: tests/test_gpu_neuralangelo.py.)"""
import numpy as np
import torch

import fixture_utils as fu
from oracle import glue_ref
from oracle import nerfacc_ref as N
from oracle import tcnn_ref as T
from test_golden_glue import binary_from, load

C5_GRID = dict(otype="HashGrid", n_levels=16, n_features_per_level=2, log2_hashmap_size=19, base_resolution=32,
               per_level_scale=1.3195079107728942)
LAMBDAS = {"lambda_rgb_l1": 1.0, "lambda_mask": 0.1, "lambda_eikonal": 0.1, "lambda_sparsity": 0.01}


def test_fixture_records_the_progressive_schedule():
    fx = load("neuralangelo_forward.npz")
    for level, step in ((4, 5), (9, 5005), (16, 12005)):
        p = f"L{level}/"
        assert int(fx[p + "global_step"]) == step
        mask = fx[p + "mask"]
        assert mask.shape == (32,) and float(mask[:2 * level].min()) == 1.0 and float(mask[2 * level:].abs().sum()) == 0.0
        # models/geometry.py:231-233: eps = 2 r / (base * scale^(level-1))
        want = 2 * 1.0 / (32 * 1.3195079107728942 ** (level - 1))
        assert abs(float(fx[p + "eps"]) - want) < 1e-12
        cn = fx[p + "gradsum/geometry.encoding.encoding.encoding.params/chunk_norms"]
        assert float(cn[level:].abs().sum()) == 0.0 and float(cn[:level].min()) > 0.0  # masked levels: zero gradient


def test_oracle_glue_matches_reference_run_of_c5():
    fx = load("neuralangelo_forward.npz")
    enc = T.Encoding(3, C5_GRID)
    sdf_mlp = torch.nn.Sequential(torch.nn.utils.weight_norm(torch.nn.Linear(35, 64)), torch.nn.Softplus(beta=100),
                                  torch.nn.utils.weight_norm(torch.nn.Linear(64, 13)))
    tex = torch.nn.Sequential(torch.nn.Linear(32, 64), torch.nn.ReLU(), torch.nn.Linear(64, 64), torch.nn.ReLU(),
                              torch.nn.Linear(64, 3))
    sh = T.Encoding(3, dict(otype="SphericalHarmonics", degree=4))
    variance = torch.tensor(float(fx["param/variance.variance"]), requires_grad=True)
    with torch.no_grad():
        enc.params.copy_(fu.seeded_normal(int(fx["table_numel"]), int(fx["table_seed"]), std=float(fx["table_std"])))
        sdf_mlp.load_state_dict({k[len("param/geometry.network.layers."):]: v for k, v in fx.items()
                                 if k.startswith("param/geometry.network.layers.")})
        tex.load_state_dict({k[len("param/texture.network.layers."):]: v for k, v in fx.items()
                             if k.startswith("param/texture.network.layers.")})
    grid = N.OccupancyGrid(fx["param/scene_aabb"], 128)
    grid._binary = binary_from(fx)
    step_size = 1.732 * 2 * 1.0 / 256
    for level in (4, 16):
        p = f"L{level}/"
        for q in list(sdf_mlp.parameters()) + list(tex.parameters()) + [enc.params, variance]:
            q.grad = None
        out = glue_ref.neus_forward(fx["rays"], enc, sdf_mlp, sh, lambda x: tex(x.float()), torch.exp(variance * 10.0),
                                    grid, fx["param/scene_aabb"], 1.0, step_size, float(fx[p + "cos_anneal_ratio"]),
                                    fx["background"], fd_eps=float(fx[p + "eps"]), level_mask=fx[p + "mask"])
        assert torch.equal(out["ray_indices"], fx[p + "out/ray_indices"])
        for k in ("sdf_samples", "sdf_grad_samples", "comp_rgb", "opacity", "depth", "weights", "comp_rgb_full"):
            assert torch.allclose(out[k], fx[p + "out/" + k], rtol=1e-5, atol=3e-6), (level, k)
        lap = fx[p + "out/sdf_laplace_samples"]
        assert torch.allclose(out["sdf_laplace_samples"], lap, rtol=1e-4, atol=1e-4 * float(lap.abs().max()))
        lam = dict(LAMBDAS, lambda_curvature=(1e-4 if level < 16 else 0.0))
        loss, terms = fu.neus_system_loss(out, fx["rgb"], fx["fg_mask"], lam)
        assert abs(float(loss) - float(fx[p + "loss"])) < 1e-5
        loss.backward()
        fu.check_grad_summary(enc.params.grad, fu.unpack_summary(fx, p + "gradsum/geometry.encoding.encoding.encoding.params"),
                              rel=2e-3, name=f"table L{level}")
        assert torch.allclose(sdf_mlp[0].weight_v.grad, fx[p + "grad/geometry.network.layers.0.weight_v"], rtol=2e-3,
                              atol=1e-6)
        assert torch.allclose(tex[4].weight.grad, fx[p + "grad/texture.network.layers.4.weight"], rtol=2e-3, atol=1e-6)
        assert torch.allclose(variance.grad, fx[p + "grad/variance.variance"], rtol=1e-3, atol=1e-6)
