"""One-kernel encode -> MLP (csrc/gridmlp.hip, tcnn.NetworkWithInputEncoding of models/network_utils.py:209-214) through the C
ABI: against the oracle's HashGrid + FullyFusedMLP restatement, and bit for bit against the two-launch HIP path (whose
parity tests are tests/test_gpu_hashgrid.py / test_gpu_mlp.py)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

# (L, F, log2T, base, per_level_scale, n_hidden, activation, n, mask_count)
CASES = [
    (16, 2, 19, 16, 1.447269237440378, 1, "none", 4099, None),    # density network of configs/nerf-blender.yaml
    (16, 2, 19, 16, 1.447269237440378, 2, "sigmoid", 1000, None),
    (16, 2, 15, 16, 1.3819, 1, "none", 5, None),                   # fewer samples than one 16-sample tile
    (16, 2, 19, 32, 1.3195, 1, "none", 2050, 9),                   # ProgressiveBandHashGrid: levels >= 9 masked
    (8, 4, 14, 16, 1.5, 2, "none", 777, None),
    (4, 8, 12, 16, 2.0, 1, "none", 333, None),
    (16, 1, 14, 16, 1.4, 1, "none", 640, None),                    # n_in = 16 < in_pad = 32: columns 16.. are the constant 1
    (12, 2, 16, 16, 1.5, 2, "none", 1234, None),                   # n_in = 24: the last lane group is half levels, half ones
]


def _setup(L, F, log2T, base, pls, n_hidden, act, n, seed=0):
    import nsr_hip
    gd = nsr_hip.make_grid_desc(L, F, log2T, base, pls)
    md = nsr_hip.NsrMlpDesc(L * F, 32, 16, 16, n_hidden, {"none": 0, "sigmoid": 1}[act])
    g = torch.Generator().manual_seed(seed)
    table = ((torch.rand(gd.n_entries * F, generator=g) * 2 - 1) * 0.5).half().cuda()
    n_w = 64 * 32 + (n_hidden - 1) * 64 * 64 + 16 * 64
    w = (torch.randn(n_w, generator=g) * 0.15).half().cuda()
    x = torch.rand(n, 3, generator=g).cuda()
    x[: min(n, 3)] = torch.tensor([[0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [0.5, 1.0, 0.0]])[: min(n, 3)].cuda()  # box faces
    return gd, md, table, w, x


@pytest.mark.parametrize("case", CASES)
def test_forward_is_bit_identical_to_encode_then_mlp(case):
    from nsr_hip import ops
    L, F, log2T, base, pls, n_hidden, act, n, mask = case
    gd, md, table, w, x = _setup(L, F, log2T, base, pls, n_hidden, act, n)
    assert ops.grid_mlp_supported(gd, md)
    enc_ref = ops.hashgrid_forward(x, table, gd, mask)
    out_ref, acts_ref = ops.mlp_forward(enc_ref, w, md, save_acts=True)
    out, acts, enc = ops.grid_mlp_forward(x, table, w, gd, md, mask, save_acts=True, want_enc=True)
    assert torch.equal(enc, enc_ref)
    assert torch.equal(acts, acts_ref)
    assert torch.equal(out, out_ref)
    out_i, acts_i, enc_i = ops.grid_mlp_forward(x, table, w, gd, md, mask)  # inference: nothing but the outputs leaves
    assert acts_i is None and enc_i is None and torch.equal(out_i, out_ref)
    _, _, enc_lm = ops.grid_mlp_forward(x, table, w, gd, md, mask, want_enc=True, enc_level_major=True)
    assert torch.equal(enc_lm.permute(1, 0, 2).reshape(n, L * F), enc_ref)


def test_forward_matches_the_oracle():
    """parity proper: oracle HashGrid (fp16 table, fp32 blend, fp16 output) -> oracle FullyFusedMLP"""
    from oracle import tcnn_ref
    from nsr_hip import ops
    L, F, log2T, base, pls, n_hidden, act, n = 16, 2, 19, 16, 1.447269237440378, 1, "none", 3001
    gd, md, table, w, x = _setup(L, F, log2T, base, pls, n_hidden, act, n, seed=5)
    og = tcnn_ref.GridDesc(L, F, log2T, base, pls)
    om = tcnn_ref.MLPDesc(L * F, 16, dict(otype="FullyFusedMLP", activation="ReLU", output_activation=act, n_neurons=64,
                                          n_hidden_layers=n_hidden))
    enc = tcnn_ref.hashgrid_encode(x.cpu(), table.float().cpu().view(-1, F), og)
    ref = tcnn_ref.mlp_forward(enc.float(), w.float().cpu(), om, return_padded=True)
    out, _, enc_hip = ops.grid_mlp_forward(x, table, w, gd, md, want_enc=True)
    # hash encode <= 1 fp16 ulp (same bound as tests/test_gpu_hashgrid.py); MLP rtol 4e-3 / atol 2e-3 (test_gpu_mlp.py)
    assert bool(((enc_hip.float().cpu() - enc.float()).abs() <= enc.float().abs() * 2 ** -10 + 1e-6).all())
    assert torch.allclose(out.float().cpu(), ref, rtol=4e-3, atol=2e-3)


def test_device_side_row_count_and_empty_input():
    import nsr_hip
    from nsr_hip import check, lib, ops, ptr, stream_ptr
    gd, md, table, w, x = _setup(16, 2, 16, 16, 1.4, 1, "none", 1000)
    out_ref, _, _ = ops.grid_mlp_forward(x, table, w, gd, md)
    live = 517
    n_dev = torch.tensor([live], dtype=torch.int32, device="cuda")
    out = torch.full((1000, 16), 7.0, dtype=torch.float16, device="cuda")
    check(lib.nsr_grid_mlp_forward(ptr(x), ptr(table), ptr(w), ptr(out), None, None, 0, 0, 1000, 16, ctypes.byref(gd),
                                   ctypes.byref(md), ptr(n_dev), stream_ptr()), "nsr_grid_mlp_forward")
    assert torch.equal(out[:live], out_ref[:live]) and bool((out[live:] == 7.0).all())  # rows behind the count: untouched
    assert lib.nsr_grid_mlp_forward(None, None, None, None, None, None, 0, 0, 0, 16, ctypes.byref(gd), ctypes.byref(md),
                                    None, stream_ptr()) == 0  # n = 0: nothing to do, nothing dereferenced


def test_unsupported_pairs_are_refused_loudly():
    import nsr_hip
    from nsr_hip import lib, ops, stream_ptr
    gd = nsr_hip.make_grid_desc(16, 4, 14, 16, 1.4)              # 64 encoded features: does not fit one k-chunk
    md = nsr_hip.NsrMlpDesc(64, 64, 16, 16, 1, 0)
    assert not ops.grid_mlp_supported(gd, md)
    x = torch.rand(8, 3, device="cuda")
    rc = lib.nsr_grid_mlp_forward(x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), None, None, 0, 0, 8, 16,
                                  ctypes.byref(gd), ctypes.byref(md), None, stream_ptr())
    assert rc != 0 and b"nsr_hashgrid_forward + nsr_mlp_forward" in lib.nsr_last_error()


def _assert_same_table_gradient(got, want, gd):
    """hashed levels accumulate in fixed point (the order of the items cannot matter): bit-identical.  The small dense
    levels sum per-thread runs and chunk slabs in fp32, in an order the binning pass's atomics decide: 1e-6."""
    F = gd.n_features
    for l in range(gd.n_levels):
        a, b = gd.offset[l] * F, gd.offset[l + 1] * F
        if gd.resolution[l] ** 3 > gd.size[l]:
            assert torch.equal(got[a:b], want[a:b]), l
        else:
            assert float((got[a:b] - want[a:b]).norm()) <= 1e-6 * float(want[a:b].norm()) + 1e-12, l


@pytest.mark.parametrize("case", [CASES[0], CASES[1], CASES[3], CASES[7]])
def test_backward_is_bit_identical_to_the_separate_calls(case):
    from nsr_hip import ops
    L, F, log2T, base, pls, n_hidden, act, n, mask = case
    gd, md, table, w, x = _setup(L, F, log2T, base, pls, n_hidden, act, n, seed=2)
    out, acts, enc = ops.grid_mlp_forward(x, table, w, gd, md, mask, save_acts=True, want_enc=True)
    dout = (torch.randn(n, 16, generator=torch.Generator().manual_seed(9)) * 1e-2).cuda()
    P = gd.n_entries * F
    gw_ref = torch.zeros(w.numel(), device="cuda")
    d_enc = ops.mlp_backward(dout, out, enc, acts, w, md, grad_weights=gw_ref, want_dx=True, grad_scale=128.0)
    gt_ref = torch.empty(P, device="cuda")
    ops.hashgrid_backward_params(x, d_enc, gt_ref, gd, mask, accumulate=False)
    gw, gt = torch.zeros(w.numel(), device="cuda"), torch.full((P,), float("nan"), device="cuda")
    ops.grid_mlp_backward(dout, out, x, enc, acts, w, gd, md, gw, gt, mask, grad_scale=128.0)
    # (the weight-gradient partials of csrc/mlp.hip meet in an order that varies from run to run: ~1e-8 apart)
    assert torch.allclose(gw, gw_ref, rtol=1e-5, atol=1e-7)
    _assert_same_table_gradient(gt, gt_ref, gd)
    assert float(gt.abs().max()) > 0
    # level-major saved encoding: the same gradients
    _, acts2, enc_lm = ops.grid_mlp_forward(x, table, w, gd, md, mask, save_acts=True, want_enc=True, enc_level_major=True)
    gw2, gt2 = torch.zeros(w.numel(), device="cuda"), torch.empty(P, device="cuda")
    ops.grid_mlp_backward(dout, out, x, enc_lm, acts2, w, gd, md, gw2, gt2, mask, grad_scale=128.0, enc_level_major=True)
    _assert_same_table_gradient(gt2, gt_ref, gd)
    assert torch.allclose(gw2, gw_ref, rtol=1e-5, atol=1e-7)


def test_network_with_input_encoding_uses_it_and_matches_the_large_batch_path(monkeypatch):
    """the drop-in module: small batches take the fused kernel, large ones the pair -- same outputs, same gradients"""
    import tinycudann as tcnn
    from nsr_hip import ops
    enc_cfg = dict(otype="HashGrid", n_levels=16, n_features_per_level=2, log2_hashmap_size=19, base_resolution=16,
                   per_level_scale=1.447269237440378)
    net_cfg = dict(otype="FullyFusedMLP", activation="ReLU", output_activation="none", n_neurons=64, n_hidden_layers=1)
    m = tcnn.NetworkWithInputEncoding(3, 16, enc_cfg, net_cfg).cuda()
    x = torch.rand(3000, 3, device="cuda")
    res = {}
    for name, thr in (("fused", 1 << 30), ("pair", 0)):
        monkeypatch.setattr(ops, "GRID_MLP_FUSED_MAX_N", thr)
        m.zero_grad(set_to_none=True)
        y = m(x)
        (y.float() ** 2).sum().backward()
        res[name] = (y.detach().clone(), m.params.grad.detach().clone())
    assert torch.equal(res["fused"][0], res["pair"][0])
    n_w = 64 * 32 + 16 * 64
    _assert_same_table_gradient(res["fused"][1][n_w:], res["pair"][1][n_w:], m.grid_desc)
    assert torch.allclose(res["fused"][1][:n_w], res["pair"][1][:n_w], rtol=1e-5, atol=1e-7)


# ---- the ray-ordered sigma pass that stops at the transmittance cut (nsr_sigma_rays) ------------------------------------------
def _ray_samples(n_rays, max_len, seed, empty_every=3):
    """ray-ordered positions + packed_info with empty rays, one-sample rays and rays of several 64-sample windows"""
    g = torch.Generator().manual_seed(seed)
    counts = torch.randint(0, max_len + 1, (n_rays,), generator=g)
    counts[::empty_every] = 0
    counts[1] = 1
    counts[2] = 64
    counts[4] = 65
    counts[5] = max_len
    starts = torch.cumsum(counts, 0) - counts
    n = int(counts.sum())
    o = torch.rand(n_rays, 3, generator=g) * 0.5 + 0.25
    dvec = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=g), dim=-1)
    ri = torch.repeat_interleave(torch.arange(n_rays), counts)
    k = torch.arange(n) - starts[ri]
    dt = 0.004
    x = (o[ri] + dvec[ri] * (k[:, None] * dt)).clamp(0, 1)
    t0 = (k * dt).float()
    t1 = t0 + dt
    packed = torch.stack([starts, counts], 1).int()
    return x.cuda().contiguous(), packed.cuda().contiguous(), t0.cuda(), t1.cuda(), n


@pytest.mark.parametrize("F,n_hidden,bias,max_len", [(2, 1, 3.0, 300), (2, 1, -1.0, 200), (2, 2, 4.0, 150), (4, 1, 3.0, 100)])
def test_sigma_rays_matches_encode_mlp_visibility_prefix_bit_for_bit(F, n_hidden, bias, max_len):
    """kept counts of every ray, and features / activations / logits of every sample in front of its ray's cut, against
    nsr_hashgrid_forward (level-major) + nsr_mlp_forward + nsr_visibility_prefix; rows behind the window that reached the cut
    stay untouched"""
    import nsr_hip
    from nsr_hip import check, lib, ptr, stream_ptr
    L = 32 // F
    gd, md, table, w, _ = _setup(L, F, 17, 16, 1.5, n_hidden, "none", 8, seed=11)
    n_rays, eps = 257, 1e-4
    x, packed, t0, t1, n = _ray_samples(n_rays, max_len, seed=3)
    C = L * F
    # reference: every marched sample
    enc_ref = torch.empty(L, n, F, dtype=torch.float16, device="cuda")
    check(lib.nsr_hashgrid_forward_ex(ptr(x), ptr(table), ptr(enc_ref), n, C, 1, L, ctypes.byref(gd), None, stream_ptr()), "enc")
    out_ref = torch.empty(n, 16, dtype=torch.float16, device="cuda")
    acts_ref = torch.empty(n_hidden, n, 64, dtype=torch.float16, device="cuda")
    check(lib.nsr_mlp_forward_ex(ptr(enc_ref), 0, C, F, ptr(w), ptr(out_ref), ptr(acts_ref), n, ctypes.byref(md), None,
                                 stream_ptr()), "mlp")
    kept_ref = torch.empty(n_rays, dtype=torch.int32, device="cuda")
    check(lib.nsr_visibility_prefix(ptr(out_ref), 16, bias, ptr(t0), ptr(t1), ptr(packed), eps, ptr(kept_ref), n_rays,
                                    stream_ptr()), "vis")
    # the ray-ordered pass into poisoned buffers
    poison = 0x7bff  # 65504: a finite half nobody computes here
    enc = torch.full((L, n, F), poison, dtype=torch.int16, device="cuda").view(torch.float16)
    out = torch.full((n, 16), poison, dtype=torch.int16, device="cuda").view(torch.float16)
    acts = torch.full((n_hidden, n, 64), poison, dtype=torch.int16, device="cuda").view(torch.float16)
    kept = torch.full((n_rays,), -7, dtype=torch.int32, device="cuda")
    check(lib.nsr_sigma_rays(ptr(x), ptr(table), ptr(w), ptr(out), ptr(acts), ptr(enc), n, ptr(packed), ptr(t0), ptr(t1),
                             bias, eps, ptr(kept), n_rays, ctypes.byref(gd), ctypes.byref(md), stream_ptr()), "sigma_rays")
    torch.cuda.synchronize()
    assert torch.equal(kept, kept_ref)
    kept_c, packed_c = kept.cpu().long(), packed.cpu().long()
    # (a low-density case keeps every sample: the pass must then evaluate everything; the first case must have real cuts)
    expect_cuts = int((kept_c + 64 <= packed_c[:, 1]).sum()) > 0
    assert 0 < int(kept_c.sum()) <= n
    if (F, n_hidden, bias) == (2, 1, 3.0):
        assert expect_cuts and int((kept_c < packed_c[:, 1]).sum()) > 10  # some rays are cut, some are not
    # rows in front of the cut: identical; rows behind the 64-sample window that reached the cut: untouched
    k = torch.arange(n) - packed_c[:, 0][torch.repeat_interleave(torch.arange(n_rays), packed_c[:, 1])]
    ray = torch.repeat_interleave(torch.arange(n_rays), packed_c[:, 1])
    front = (k < kept_c[ray]).cuda()
    window_end = torch.minimum(packed_c[:, 1], (kept_c // 64 + 1) * 64)  # (a ray whose cut falls on a window edge goes one further)
    behind = (k >= window_end[ray]).cuda()
    assert torch.equal(out[front], out_ref[front])
    assert torch.equal(enc[:, front], enc_ref[:, front])
    assert torch.equal(acts[:, front], acts_ref[:, front])
    assert bool((out[behind].view(torch.int16) == poison).all()) and bool((enc[:, behind].view(torch.int16) == poison).all())
    if expect_cuts:
        assert int(behind.sum()) > 0 and float((~behind).float().mean()) < 0.95
    else:
        assert int(behind.sum()) == 0


def test_prune_pass_sigma_modes_agree():
    """nsr_nerf_prune_pass with the ray-ordered sigma pass vs the three stand-alone launches, through the fused step:
    identical kept counts / packing, identical rendering"""
    import nsr
    from nsr_hip import lib
    from nsr.fused import FusedNeRFStep
    torch.manual_seed(0)
    cfg = nsr.configs.get("nerf-blender")
    model = nsr.build(cfg).cuda().train()
    with torch.no_grad():
        model.geometry.encoding_with_network.params[3072:].normal_(0, 0.4)  # dense enough for cuts inside the rays
    model.occupancy_grid._binary[:] = True
    model.randomized = False
    g = torch.Generator().manual_seed(1)
    o = torch.tensor([[0.0, 0.0, 4.0]]).repeat(512, 1)
    dvec = torch.nn.functional.normalize(torch.randn(512, 3, generator=g) * 0.2 + torch.tensor([0.0, 0.0, -1.0]), dim=-1)
    rays = torch.cat([o, dvec], -1).cuda()
    res = {}
    try:
        for mode in (0, 1):
            lib.nsr_nerf_sigma_mode(mode)
            step = FusedNeRFStep(model)
            model.zero_grad(set_to_none=True)
            out = step.forward_backward(rays, torch.full((512, 3), 0.5, device="cuda"), torch.ones(3, device="cuda"))
            res[mode] = (int(out["num_samples"]), out["comp_rgb"].clone(), out["ray_indices"].clone(),
                         model.geometry.encoding_with_network.params.grad[:3072].clone())
    finally:
        lib.nsr_nerf_sigma_mode(0)
    assert res[0][0] == res[1][0] > 1000
    assert torch.equal(res[0][2], res[1][2])
    assert torch.equal(res[0][1], res[1][1])
