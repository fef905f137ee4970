"""SURVEY.md 8f row 4 on the MI355X: (a) the product state holder has EXACTLY the reference's state-dict keys and shapes
(tests/golden/state_dict_keys.json, minted by tests/gen_state_keys.py from models.make on the real YAMLs) and round-trips a Lightning-style
checkpoint; (b) eval-mode chunked rendering (models/utils.py:13-50 semantics) equals the one-shot forward; (c) the
isosurface lattice evaluation (models/geometry.py:83-100) equals the level function called directly."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["nerf-blender", "neus-blender", "neus-dtu", "neuralangelo"])
def test_state_dict_keys_match_reference_and_checkpoint_round_trips(name, tmp_path):
    import nsr
    want = json.load(open(os.path.join(GOLD, "state_dict_keys.json")))[name]
    st = nsr.build(nsr.configs.get(name)).cuda()
    sd = st.state_dict()
    assert sorted(sd) == sorted(want), set(sd) ^ set(want)
    for k, shape in want.items():
        assert list(sd[k].shape) == shape, (k, list(sd[k].shape), shape)
    # a Lightning checkpoint nests the model under "model." (systems/base.py: self.model); occupancy grid included
    with torch.no_grad():
        for p in st.parameters():
            if p.numel():
                p.normal_(0, 0.01)
        st.occupancy_grid._binary[10:20, 30:40, 50:60] = True
    ckpt = {"state_dict": {"model." + k: v.detach().cpu().clone() for k, v in st.state_dict().items()}, "global_step": 20000}
    path = tmp_path / "epoch=0-step=20000.ckpt"
    torch.save(ckpt, path)
    st2 = nsr.build(nsr.configs.get(name)).cuda()
    missing, unexpected = st2.load_reference_checkpoint(torch.load(path)["state_dict"])
    assert not missing and not unexpected
    for k, v in st.state_dict().items():
        assert torch.equal(st2.state_dict()[k], v), k
    assert int(st2.occupancy_grid.binary.sum()) == 1000


def _rays(n, seed=0, r=4.0):
    g = torch.Generator().manual_seed(seed)
    o = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1) * r
    d = torch.nn.functional.normalize(-o + torch.randn(n, 3, generator=g) * 0.4, dim=-1)
    return torch.cat([o, d], -1).cuda()


def test_eval_chunked_render_nerf_and_neus():
    import nsr
    from nsr.export import render_rays
    from nsr.fused import FusedNeRFStep
    from nsr.fused_neus import FusedNeuSStep
    torch.manual_seed(0)
    st = nsr.build(nsr.configs.get("nerf-blender")).cuda().eval()
    with torch.no_grad():
        st.geometry.encoding_with_network.params[3072:].normal_(0, 0.1)
    st.occupancy_grid._binary[24:104, 24:104, 24:104] = True
    run = FusedNeRFStep(st)
    rays = _rays(1000)
    whole = render_rays(run, rays, chunk=4096, move_to_cpu=False)
    parts = render_rays(run, rays, chunk=333)
    assert not parts["comp_rgb"].is_cuda and parts["comp_rgb"].shape == (1000, 3)
    for k in ("comp_rgb", "opacity", "depth"):
        assert torch.allclose(parts[k], whole[k].cpu(), atol=1e-6), k
    assert int(parts["num_samples"].sum()) == int(whole["num_samples"].sum()) and parts["num_samples"].numel() == 4
    assert st.randomized is False

    cfg = nsr.configs.get("neus-blender")
    sn = nsr.build(cfg).cuda().eval()
    sn.update_step(0, 1000)
    sn.occupancy_grid._binary[24:104, 24:104, 24:104] = True
    rn = FusedNeuSStep(sn)
    whole = render_rays(rn, rays, chunk=4096, move_to_cpu=False)
    parts = render_rays(rn, rays, chunk=400)
    for k in ("comp_rgb_full", "opacity", "depth", "comp_normal"):
        assert torch.allclose(parts[k], whole[k].cpu(), atol=1e-6), k
    assert all(p.grad is None for p in sn.parameters())


def test_isosurface_lattice_matches_direct_level_evaluation():
    import nsr
    from nsr.export import forward_level, isosurface_levels
    for name in ("nerf-blender", "neus-blender"):
        st = nsr.build(nsr.configs.get(name)).cuda().eval()
        with torch.no_grad():
            for p in st.parameters():
                if p.numel() > 100000:
                    p.normal_(0, 0.05)
        st.update_step(0, 0)
        res = 40
        vol = isosurface_levels(st, res, chunk=7 * res * res)  # several ragged chunks
        assert vol.shape == (res, res, res) and not vol.is_cuda
        r = float(st.config["radius"])
        lin = torch.linspace(0, 1, res)
        gx, gy, gz = torch.meshgrid(lin, lin, lin, indexing="ij")
        pts = (torch.stack([gx, gy, gz], -1).reshape(-1, 3) * 2 * r - r).cuda()
        want = forward_level(st, pts).float().cpu().view(res, res, res)
        assert torch.allclose(vol, want, atol=1e-6)
        if name == "neus-blender":  # sphere initialisation: the level set is (about) the sphere of radius 0.5
            c = res // 2
            assert float(vol[c, c, c]) < 0 < float(vol[0, 0, 0])


def test_vertex_colors_match_the_reference_export_queries():
    """nsr.export.vertex_colors == the colour queries of the reference's export() (models/nerf.py:152-161: viewing direction
    -z; models/neus.py:313-323: texture(feature, -normal, normal)) evaluated through the reference-shaped modules"""
    import nsr
    import refmirror
    from nsr.export import vertex_colors
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(3)
    # NeRF
    m = refmirror.NeRFModel(nsr.configs.get("nerf-blender")).cuda().eval()
    with torch.no_grad():
        m.geometry.encoding_with_network.params[3072:].normal_(0, 0.1)
    pts = ((torch.rand(5000, 3, generator=g) * 2 - 1) * 1.4).cuda()
    with torch.no_grad():
        _, feature = m.geometry(pts)
        dirs = torch.zeros_like(pts)
        dirs[:, 2] = -1.0
        want = m.texture(feature, dirs).clamp(0, 1)
    got = vertex_colors(m, pts, chunk=1777)
    assert not got.is_cuda and got.shape == (5000, 3)
    assert torch.allclose(got, want.cpu(), atol=3e-3), float((got - want.cpu()).abs().max())
    # NeuS (analytic normals, fused colour MLP) and neuralangelo (finite differences, fp32 colour MLP)
    for name, step in (("neus-blender", 1000), ("neuralangelo", 12005)):
        cfg = nsr.configs.get(name)
        m = refmirror.NeuSModel(cfg).cuda().eval()
        with torch.no_grad():
            enc = m.geometry.encoding.encoding
            (enc.encoding if hasattr(enc, "encoding") else enc).params.normal_(0, 0.05)
            m.geometry.network.layers[0].weight_v[:, 3:].normal_(0, 0.05)
        m.update_step(0, step)
        r = float(cfg["radius"])
        pts = ((torch.rand(4000, 3, generator=g) * 2 - 1) * 0.95 * r).cuda()
        fd = cfg["geometry"]["grad_type"] == "finite_difference"
        with torch.enable_grad():
            outs = m.geometry(pts, with_grad=True, with_feature=True, **({"with_laplace": True} if fd else {}))
        sdf, sdf_grad, feature = outs[0], outs[1], outs[2]
        normal = torch.nn.functional.normalize(sdf_grad.detach(), p=2, dim=-1)
        with torch.no_grad():
            want = m.texture(feature.detach(), -normal, normal)
        got = vertex_colors(m, pts, chunk=1500)
        bad = float(((got - want.cpu()).abs() > 5e-3).float().mean())  # (samples on a fine-level cell boundary: see DESIGN 2)
        assert bad < 5e-3, (name, bad, float((got - want.cpu()).abs().max()))
        at = m._level_runner.surface_attributes(pts[:1000])
        assert torch.allclose(at["sdf"], sdf[:1000].detach().view(-1), atol=1e-3)


def test_export_after_checkpoint_load_needs_the_schedules_of_its_global_step():
    """a progressive / finite-difference model restored from a checkpoint has no active level count and no eps until the
    step-dependent schedules are restored (the reference does it in its batch-start hooks from the checkpoint's
    global_step): exporting without them is refused, `load_reference_checkpoint(ckpt)` with the Lightning checkpoint's
    `global_step` restores them"""
    import nsr
    from nsr.export import isosurface_levels
    cfg = nsr.configs.get("neuralangelo")
    src = nsr.build(cfg).cuda().train()
    src.update_step(0, 9005)                      # level 4 + 9 = 13 active
    ckpt = {"state_dict": {"model." + k: v for k, v in src.state_dict().items()}, "global_step": 9005, "epoch": 0}
    dst = nsr.build(cfg).cuda().eval()
    dst.load_reference_checkpoint(ckpt["state_dict"])
    with pytest.raises(RuntimeError, match="restore_schedules"):
        isosurface_levels(dst, 16)
    dst2 = nsr.build(cfg).cuda().eval()
    dst2.load_reference_checkpoint(ckpt)          # the whole Lightning checkpoint: state_dict + global_step
    assert dst2.current_level == src.current_level == 13
    assert abs(dst2.finite_difference_eps - src.finite_difference_eps) < 1e-12
    a, b = isosurface_levels(dst2, 16), isosurface_levels(src.eval(), 16)
    assert torch.equal(a, b)
