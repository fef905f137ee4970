"""HIP fully-fused MLP (MFMA) vs the oracle's fp16-emulating restatement, through the C ABI."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk(n_in, n_out, n_hidden, act, n, seed=0, x_f32=True):
    from oracle import tcnn_ref
    import nsr_hip
    cfg = dict(otype="FullyFusedMLP", activation="ReLU", output_activation=act, n_neurons=64, n_hidden_layers=n_hidden)
    od = tcnn_ref.MLPDesc(n_in, n_out, cfg)
    hd = nsr_hip.make_mlp_desc(n_in, n_out, n_hidden, act)
    g = torch.Generator().manual_seed(seed)
    params = tcnn_ref.init_mlp_params(od, seed).half().float()  # asymmetric random weights (transpose-detecting)
    x = torch.randn(n, n_in, generator=g)
    if not x_f32:
        x = x.half().float()
    return od, hd, params, x


@pytest.mark.parametrize("n_in,n_out,n_hidden,act,n", [
    (32, 16, 1, "none", 1000),      # density MLP of configs/nerf-blender.yaml:50-55
    (32, 3, 2, "sigmoid", 777),     # colour MLP  nerf-blender.yaml:62-67 / neus-blender.yaml:70-75
    (16, 3, 2, "none", 64),
    (19, 13, 3, "none", 130),       # padded inputs (constant 1.0) and 48-wide first layer
    (64, 16, 4, "sigmoid", 33),
    (32, 16, 1, "none", 5),         # ragged: fewer samples than one 16-sample tile
])
def test_forward_parity(n_in, n_out, n_hidden, act, n):
    from oracle import tcnn_ref
    from nsr_hip import ops
    od, hd, params, x = _mk(n_in, n_out, n_hidden, act, n)
    ref = tcnn_ref.mlp_forward(x, params, od, return_padded=True)
    out, acts = ops.mlp_forward(x.cuda(), params.half().cuda(), hd, save_acts=True)
    out = out.float().cpu()
    # fp16 activations + fp32 accumulate on both sides; MFMA sums in a different order
    assert torch.allclose(out[:, :n_out], ref[:, :n_out], rtol=4e-3, atol=2e-3), (out - ref)[:, :n_out].abs().max()
    assert acts.shape == (n_hidden, n, 64) and bool((acts >= 0).all())


def test_forward_fp16_input_fast_path_matches_fp32_input():
    from nsr_hip import ops
    od, hd, params, x = _mk(32, 16, 1, "none", 4096, x_f32=False)
    w = params.half().cuda()
    a, _ = ops.mlp_forward(x.cuda(), w, hd, save_acts=False)
    b, _ = ops.mlp_forward(x.half().cuda(), w, hd, save_acts=False)
    assert torch.equal(a, b)


@pytest.mark.parametrize("n_in,n_out,n_hidden,act,n", [
    (32, 16, 1, "none", 2049),
    (32, 3, 2, "sigmoid", 1500),
    (19, 13, 3, "none", 300),
    (32, 3, 2, "none", 7),
])
def test_backward_parity(n_in, n_out, n_hidden, act, n):
    from oracle import tcnn_ref
    from nsr_hip import ops
    od, hd, params, x = _mk(n_in, n_out, n_hidden, act, n, seed=3)
    g = torch.Generator().manual_seed(11)
    dout = torch.randn(n, n_out, generator=g) * 0.01
    p = params.clone().requires_grad_(True)
    xo = x.clone().requires_grad_(True)
    tcnn_ref.mlp_forward(xo, p, od, fp16=False).backward(dout)  # fp32 reference gradients
    w = params.half().cuda()
    out, acts = ops.mlp_forward(x.cuda(), w, hd, save_acts=True)
    gw = torch.zeros(od.n_params, device="cuda")
    dx = ops.mlp_backward(dout.cuda(), out, x.cuda(), acts, w, hd, grad_weights=gw, want_dx=True, grad_scale=128.0)
    gw, dx = gw.cpu(), dx.cpu()
    # rows of W_last beyond n_out and columns of W0 that multiply nothing get zero grad in both
    ref_w, ref_x = p.grad, xo.grad
    cos_w = torch.nn.functional.cosine_similarity(gw, ref_w, dim=0)
    cos_x = torch.nn.functional.cosine_similarity(dx.flatten(), ref_x.flatten(), dim=0)
    assert cos_w > 0.999 and cos_x > 0.999, (cos_w, cos_x)
    import fixture_utils as fu
    # rel-L2 <= 1e-2 and cosine >= 0.999.  The reference here is FP32 autograd; the kernel (like tcnn's) carries fp16
    # activations and an fp16 gradient chain: through one hidden layer it measures 4-8e-3, through two or three 1.2-1.8e-2
    # (profiles/r05_grad_parity.json) -- the fp16 chain's floor against an fp32 reference, so those shapes take 2e-2 (cosine
    # still >= 0.999: measured 0.99985-0.99993); the fused paths, compared with fp16-emulating fixtures, all meet 1e-2
    rel = 1e-2 if n_hidden == 1 else 2e-2
    floor = None if n_hidden == 1 else ((1.2e-2, 1.8e-2), "fp16 activations + fp16 gradient chain through >= 2 hidden layers "
                                                            "against FP32 autograd (tcnn has the same; profiles/r05_grad_parity.json)")
    fu.assert_grad(gw, ref_w, ("weights", n_in, n_out, n_hidden, act), rel=rel, floor=floor)
    fu.assert_grad(dx, ref_x, ("input", n_in, n_out, n_hidden, act), rel=rel, floor=floor)


def test_backward_accumulates_into_grad():
    from nsr_hip import ops
    od, hd, params, x = _mk(32, 16, 1, "none", 512)
    w = params.half().cuda()
    out, acts = ops.mlp_forward(x.cuda(), w, hd, save_acts=True)
    dout = torch.randn(512, 16, device="cuda") * 0.01
    g1 = torch.zeros(od.n_params, device="cuda")
    ops.mlp_backward(dout, out, x.cuda(), acts, w, hd, grad_weights=g1)
    g2 = g1.clone()
    ops.mlp_backward(dout, out, x.cuda(), acts, w, hd, grad_weights=g2)
    assert torch.allclose(g2, 2 * g1, rtol=1e-5, atol=1e-7)  # deterministic partial-sum reduction
