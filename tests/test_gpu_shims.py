"""The drop-in ``tinycudann`` / ``nerfacc`` packages vs the oracle's stand-ins, module against module."""
import pytest
import torch

from conftest import COLOR_MLP, DENSITY_MLP, NERF_GRID, NEUS_GRID

pytestmark = pytest.mark.gpu


def _copy_params(dst, src):
    with torch.no_grad():
        dst.params.copy_(src.params.detach().to(dst.params.device))


def test_network_with_input_encoding_fwd_bwd():
    import tinycudann as tcnn
    from oracle import tcnn_ref
    ref = tcnn_ref.NetworkWithInputEncoding(3, 16, NERF_GRID, DENSITY_MLP)
    with torch.no_grad():  # a "trained-like" table so the output is not ~0
        ref.params[ref.desc.n_params:] = torch.randn(ref.grid.n_params) * 0.1
    net = tcnn.NetworkWithInputEncoding(n_input_dims=3, n_output_dims=16, encoding_config=NERF_GRID,
                                        network_config=DENSITY_MLP)
    assert [n for n, _ in net.named_parameters()] == ["params"]
    assert net.params.dtype == torch.float32 and net.params.numel() == 12602992 == ref.params.numel()
    assert net.n_input_dims == 3 and net.n_output_dims == 16 and net.loss_scale == 128.0
    _copy_params(net, ref)
    x = torch.rand(3000, 3)
    y_ref = ref(x)
    y = net(x.cuda())
    assert y.dtype == torch.float16 and y.shape == (3000, 16)
    assert torch.allclose(y.float().cpu(), y_ref.float(), rtol=1e-2, atol=3e-3)
    g = torch.randn(3000, 16) * 0.01
    y_ref.float().backward(g)
    y.float().backward(g.cuda())
    gr, gg = ref.params.grad, net.params.grad.cpu()
    n_net = ref.desc.n_params
    for a, b in ((gr[:n_net], gg[:n_net]), (gr[n_net:], gg[n_net:])):
        assert torch.nn.functional.cosine_similarity(a, b, dim=0) > 0.999
        assert (a - b).norm() / a.norm() < 3e-2


def test_encoding_composite_double_backward_like_volume_sdf():
    """the autograd pattern of reference models/geometry.py:161-180 + eikonal loss (systems/neus.py:106)"""
    import tinycudann as tcnn
    from oracle import tcnn_ref
    ref = tcnn_ref.Encoding(3, NEUS_GRID)
    with torch.no_grad():
        ref.params.copy_(torch.randn(ref.desc.n_params) * 0.05)
    enc = tcnn.Encoding(3, NEUS_GRID)
    assert enc.n_output_dims == 32 and enc.params.numel() == 13969152
    _copy_params(enc, ref)
    lin = torch.nn.Linear(35, 1)
    x0 = torch.rand(800, 3)

    def run(e, dev):
        x = x0.clone().to(dev).detach().requires_grad_(True)
        l = torch.nn.Linear(35, 1).to(dev)
        l.load_state_dict(lin.state_dict())
        feat = torch.cat([x * 2 - 1, e(x)], dim=-1)
        sdf = l(feat.float())[..., 0]
        (grad,) = torch.autograd.grad(sdf, x, torch.ones_like(sdf), create_graph=True)
        loss = ((grad.norm(dim=-1) - 1) ** 2).mean() + sdf.mean()
        loss.backward()
        return grad.detach().cpu(), e.params.grad.cpu(), l.weight.grad.cpu(), x.grad.cpu()

    g_ref, pg_ref, lw_ref, xg_ref = run(ref, "cpu")
    g, pg, lw, xg = run(enc, "cuda")
    assert (g - g_ref).norm() / g_ref.norm() < 5e-3
    assert (pg - pg_ref).norm() / pg_ref.norm() < 2e-2
    assert (lw - lw_ref).norm() / lw_ref.norm() < 1e-2
    assert (xg - xg_ref).norm() / xg_ref.norm() < 2e-2


def test_sh_and_network_like_volume_radiance():
    import tinycudann as tcnn
    from oracle import tcnn_ref
    sh_ref, sh = tcnn_ref.Encoding(3, dict(otype="SphericalHarmonics", degree=4)), tcnn.Encoding(
        3, dict(otype="SphericalHarmonics", degree=4))
    assert sh.n_output_dims == 16 and sh.params.numel() == 0
    net_ref, net = tcnn_ref.Network(32, 3, COLOR_MLP), tcnn.Network(32, 3, COLOR_MLP)
    assert net.params.numel() == 7168
    _copy_params(net, net_ref)
    d = torch.nn.functional.normalize(torch.randn(2000, 3), dim=-1)
    feat = torch.randn(2000, 16)
    e_ref, e = sh_ref((d + 1) / 2), sh(((d + 1) / 2).cuda())
    assert torch.allclose(e.float().cpu(), e_ref.float(), rtol=2e-3, atol=1e-3)
    out_ref = net_ref(torch.cat([feat, e_ref], -1))
    out = net(torch.cat([feat.cuda(), e], -1))
    assert out.shape == (2000, 3) and out.dtype == torch.float16
    assert torch.allclose(out.float().cpu(), out_ref.float(), rtol=5e-3, atol=3e-3)


def test_unknown_keys_ignored_and_unsupported_raise():
    import tinycudann as tcnn
    cfg = dict(NEUS_GRID, include_xyz=True, start_level=4, start_step=0, update_steps=1000)  # leaked keys
    assert tcnn.Encoding(3, cfg).n_output_dims == 32
    tcnn.Network(32, 3, dict(COLOR_MLP, sphere_init=True, weight_norm=True))
    with pytest.raises(NotImplementedError):
        tcnn.Encoding(3, dict(otype="Frequency", n_frequencies=6))
    with pytest.raises(NotImplementedError):
        tcnn.Network(32, 3, dict(COLOR_MLP, n_neurons=128))
    with pytest.raises(Exception):
        tcnn.Encoding(3, NERF_GRID)(torch.rand(4, 3))  # CPU input: no silent fallback
    tcnn.free_temporary_memory()


def test_ray_marching_with_sigma_fn_matches_oracle():
    import nerfacc as A
    from oracle import nerfacc_ref as N
    from test_gpu_march import _scene
    o, d, roi, binary = _scene(500, seed=4)
    gref = N.OccupancyGrid(roi, 128)
    gref._binary = binary.clone()
    ggpu = A.OccupancyGrid(roi, 128).cuda()
    ggpu._binary = binary.cuda()
    assert set(ggpu.state_dict().keys()) == {"_roi_aabb", "_binary", "resolution", "occs"} == set(gref.state_dict().keys())

    def sigma(t0, t1, ri):  # analytic density so both sides agree to fp32 rounding
        dev = t0.device
        p = o.to(dev)[ri.long()] + d.to(dev)[ri.long()] * (t0 + t1) / 2
        return 200.0 * torch.exp(-4 * (p ** 2).sum(-1, keepdim=True))

    step = 1.732 * 2 * 1.5 / 1024
    ri_ref, t0_ref, t1_ref = N.ray_marching(o, d, scene_aabb=roi, grid=gref, sigma_fn=sigma, render_step_size=step)
    ri, t0, t1 = A.ray_marching(o.cuda(), d.cuda(), scene_aabb=roi.cuda(), grid=ggpu, sigma_fn=sigma,
                                render_step_size=step)
    assert ri.dtype == torch.int64 and t0.shape[1] == 1
    # visibility pruning compares T against 1e-4: identical except samples whose T is within rounding of it
    assert abs(ri.numel() - ri_ref.numel()) <= max(2, ri_ref.numel() // 10000)
    if ri.numel() == ri_ref.numel():
        assert torch.equal(ri.cpu(), ri_ref) and torch.equal(t0.cpu(), t0_ref)
    # without pruning: bit exact
    ri_ref, t0_ref, t1_ref = N.ray_marching(o, d, scene_aabb=roi, grid=gref, render_step_size=step)
    ri, t0, t1 = A.ray_marching(o.cuda(), d.cuda(), scene_aabb=roi.cuda(), grid=ggpu, render_step_size=step)
    assert torch.equal(ri.cpu(), ri_ref) and torch.equal(t0.cpu(), t0_ref) and torch.equal(t1.cpu(), t1_ref)
    # tensor near_plane (reference models/neus.py:157,164) and float far_plane
    near = torch.rand(500) * 3
    ri_ref, t0_ref, _ = N.ray_marching(o, d, grid=gref, near_plane=near, far_plane=6.0, render_step_size=step)
    ri, t0, _ = A.ray_marching(o.cuda(), d.cuda(), grid=ggpu, near_plane=near.cuda(), far_plane=6.0,
                               render_step_size=step)
    assert torch.equal(ri.cpu(), ri_ref) and torch.equal(t0.cpu(), t0_ref)


def test_occupancy_grid_update_cells_matches_oracle():
    import nerfacc as A
    from oracle import nerfacc_ref as N
    roi = torch.tensor([-1.5] * 3 + [1.5] * 3)
    for ctype_ref, ctype in ((N.ContractionType.AABB, A.ContractionType.AABB),
                             (N.ContractionType.UN_BOUNDED_SPHERE, A.ContractionType.UN_BOUNDED_SPHERE)):
        gref, ggpu = N.OccupancyGrid(roi, 32, ctype_ref), A.OccupancyGrid(roi, 32, ctype).cuda()
        gref.train(), ggpu.train()
        idx = torch.randperm(32 ** 3)[:20000]  # unique cells: duplicate scatter targets are order-dependent
        jit = torch.rand(20000, 3)
        fn = lambda x: torch.exp(-2 * (x ** 2).sum(-1, keepdim=True)) * 0.05  # noqa: E731
        gref._update_cells(idx, jit, fn, occ_thre=0.01)
        ggpu._update_cells(idx.cuda(), jit.cuda(), fn, occ_thre=0.01)
        assert torch.allclose(ggpu.occs.cpu(), gref.occs, rtol=1e-5, atol=1e-7)
        assert float((ggpu.binary.cpu() != gref.binary).float().mean()) < 1e-4
    ggpu.eval()
    with pytest.raises(RuntimeError):
        ggpu.every_n_step(step=0, occ_eval_fn=fn)


@pytest.mark.gpu
def test_half_transport_casts_round_trip():
    """nsr_scale_to_half / nsr_scale_from_half (gradient transport of nsr/parallel.py)"""
    from nsr_hip import check, lib, ptr, stream_ptr
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(1_000_003, device="cuda", generator=g) * 1e-3
    h = torch.empty(x.numel(), dtype=torch.float16, device="cuda")
    y = torch.empty_like(x)
    check(lib.nsr_scale_to_half(ptr(x), ptr(h), x.numel(), 1024.0, stream_ptr()), "nsr_scale_to_half")
    check(lib.nsr_scale_from_half(ptr(h), ptr(y), x.numel(), 1.0 / 2048.0, stream_ptr()), "nsr_scale_from_half")
    assert torch.equal(h, (x * 1024.0).half())
    assert torch.equal(y, h.float() / 2048.0)


def _reference_sphere_init(n_input_dims, n_output_dims, config, network):
    """the body of the reference's sphere_init_tcnn_network (models/network_utils.py:142-173): the flat parameter is
    re-written THROUGH ``.data`` (no version bump) after asserting its length -- the layout contract of a tcnn.Network"""
    import math
    padto = 16 if config["otype"] == "FullyFusedMLP" else 8
    n_input_dims = n_input_dims + (padto - n_input_dims % padto) % padto
    n_output_dims = n_output_dims + (padto - n_output_dims % padto) % padto
    data = list(network.parameters())[0].data
    assert data.shape[0] == (n_input_dims + n_output_dims) * config["n_neurons"] + (config["n_hidden_layers"] - 1) * config["n_neurons"] ** 2
    new_data = []
    weight = torch.zeros((config["n_neurons"], n_input_dims)).to(data)
    torch.nn.init.constant_(weight[:, 3:], 0.0)
    torch.nn.init.normal_(weight[:, :3], 0.0, math.sqrt(2) / math.sqrt(config["n_neurons"]))
    new_data.append(weight.flatten())
    for _ in range(config["n_hidden_layers"] - 1):
        weight = torch.zeros((config["n_neurons"], config["n_neurons"])).to(data)
        torch.nn.init.normal_(weight, 0.0, math.sqrt(2) / math.sqrt(config["n_neurons"]))
        new_data.append(weight.flatten())
    weight = torch.zeros((n_output_dims, config["n_neurons"])).to(data)
    torch.nn.init.normal_(weight, mean=math.sqrt(math.pi) / math.sqrt(config["n_neurons"]), std=0.0001)
    new_data.append(weight.flatten())
    new_data = torch.cat(new_data)
    data.copy_(new_data)
    return new_data


@pytest.mark.parametrize("warm", ["fresh", "after_no_grad_forward", "after_training_forward"])
def test_sphere_init_through_params_data_is_seen_and_trains(warm):
    """get_mlp(..., sphere_init=True) of the reference (models/network_utils.py:176-184) on the HIP tcnn.Network: the
    ``.data`` write must reach the fp16 image the kernels read -- also when a forward already cached one -- the network then
    represents the sphere |x| - r, and one optimizer step trains it"""
    import tinycudann as tcnn
    from oracle import tcnn_ref
    cfg = dict(otype="FullyFusedMLP", activation="ReLU", output_activation="none", n_neurons=64, n_hidden_layers=2,
               sphere_init=True)  # (unknown keys are ignored, like tcnn does)
    torch.manual_seed(3)
    net = tcnn.Network(3, 13, cfg).train()
    x = (torch.rand(2048, 3, device="cuda") - 0.5) * 2
    if warm == "after_no_grad_forward":
        with torch.no_grad():
            net(x)
    elif warm == "after_training_forward":
        net(x).float().sum().backward()
        net.params.grad = None
    new_data = _reference_sphere_init(3, 13, cfg, net)
    assert torch.equal(net.params.data, new_data)
    want = tcnn_ref.mlp_forward(x.cpu(), new_data.cpu(), tcnn_ref.MLPDesc(3, 13, cfg))
    for grad_mode in (False, True):
        with torch.set_grad_enabled(grad_mode):
            y = net(x)
        assert torch.allclose(y.float().cpu(), want.float(), rtol=1e-2, atol=3e-3), (warm, grad_mode)
    # the geometric init: output 0 grows with |x| (a sphere's signed distance up to scale and offset; bias-free fused MLP, so
    # only approximately -- the strict check is the oracle forward above)
    r = x.norm(dim=-1)
    corr = torch.corrcoef(torch.stack([y[:, 0].float(), r]))[0, 1]
    assert float(corr) > 0.8, float(corr)
    # ... and it trains: one AdamW step on the eikonal-free target  out[:, 0] -> |x| - 0.5
    opt = torch.optim.AdamW(net.parameters(), lr=1e-2)
    losses = []
    for _ in range(30):
        loss = ((net(x)[:, 0].float() - (r - 0.5)) ** 2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < 0.5 * losses[0], losses[::6]
    assert not torch.equal(net.params.data, new_data)
