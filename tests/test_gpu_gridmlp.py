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
