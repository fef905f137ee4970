"""csrc/vmlp.hip (fp32 VanillaMLP on f32 MFMA) against plain PyTorch fp32 autograd of the same nn.Linear stack
(reference models/network_utils.py:95-139), including the analytic-normal protocol of models/geometry.py:176-180:
g = d out[0] / d input with create_graph, then a loss on g (double backward).  Tolerances: forward rtol 1e-5,
gradients rel-L2 1e-4 (fp32 both sides, different summation order)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.double() - b.double()).norm() / max(float(b.double().norm()), 1e-30))


def _net(n_in, n_out, nh, softplus, weight_norm, seed):
    torch.manual_seed(seed)
    act = (lambda: torch.nn.Softplus(beta=100)) if softplus else (lambda: torch.nn.ReLU())
    dims = [n_in] + [64] * nh + [n_out]
    mods = []
    for i in range(len(dims) - 1):
        lin = torch.nn.Linear(dims[i], dims[i + 1])
        with torch.no_grad():
            lin.bias.normal_(0, 0.1)
            if softplus:
                lin.weight.mul_(0.3)  # keep 100 z inside softplus's curved range for a good share of the units
        mods.append(torch.nn.utils.weight_norm(lin) if weight_norm else lin)
        if i < len(dims) - 2:
            mods.append(act())
    return torch.nn.Sequential(*mods).cuda()


def _linears(net):
    return [m for m in net if isinstance(m, torch.nn.Linear)]


@pytest.mark.parametrize("n", [1, 1000, 4099])
@pytest.mark.parametrize("n_in,n_out,nh,softplus,wn", [(32, 3, 2, False, False), (35, 13, 1, True, True),
                                                        (24, 3, 2, False, False), (32, 8, 1, False, False)])
def test_forward_backward_match_torch(n, n_in, n_out, nh, softplus, wn):
    from nsr.fused_neus import VanillaBlob
    from nsr_hip import check, lib, ptr, stream_ptr
    net = _net(n_in, n_out, nh, softplus, wn, seed=n_in + nh)
    vb = VanillaBlob(_linears(net), n_in, n_out, activation=int(softplus))
    blob = vb.build()
    g = torch.Generator(device="cuda").manual_seed(n)
    x = torch.randn(n, n_in, device="cuda", generator=g)
    d_out = torch.randn(n, n_out, device="cuda", generator=g)
    xr = x.clone().requires_grad_(True)
    want = net(xr)
    (want * d_out).sum().backward()
    out = torch.empty(n, 16, device="cuda")
    d, s = vb.desc, stream_ptr()
    check(lib.nsr_vmlp_forward(ctypes.byref(d), ptr(blob.detach()), ptr(x), n_in, None, 0, ptr(out), None, None, n, n, None,
                               s), "fwd")
    assert torch.allclose(out[:, :n_out], want.detach(), rtol=1e-5, atol=1e-5), float((out[:, :n_out] - want).abs().max())
    assert float(out[:, n_out:].abs().max() if n_out < 16 else 0.0) == 0.0
    d16 = torch.zeros(n, 16, device="cuda")
    d16[:, :n_out] = d_out
    d_x = torch.empty(n, n_in, device="cuda")
    gb = torch.empty(vb.n_floats, device="cuda")
    ws = torch.empty(int(lib.nsr_vmlp_backward_workspace_floats(ctypes.byref(d), n)), device="cuda")
    check(lib.nsr_vmlp_backward(ctypes.byref(d), ptr(blob.detach()), ptr(x), n_in, None, 0, ptr(d16), None, None, ptr(d_x),
                                n_in, 0, n_in, 0, ptr(gb), 0, ptr(ws), n, n, None, s), "bwd")
    assert _rel(d_x, xr.grad) < 1e-4, _rel(d_x, xr.grad)
    want_grads = [p.grad.clone() for p in net.parameters()]
    net.zero_grad()
    vb.push_gradient(gb)
    for (k, p), w in zip(net.named_parameters(), want_grads):
        assert _rel(p.grad, w) < 2e-4, (k, _rel(p.grad, w))


@pytest.mark.parametrize("n,n_taps", [(777, 0), (500, 6)])
def test_sdf_input_mode_analytic_normal_and_double_backward(n, n_taps):
    """input = [2 x - 1 | fp16 encoding]; g_in = d sdf / d input; loss = <d_out, out> + <P, g_in> (+ taps: column 0 only)"""
    from nsr.fused_neus import VanillaBlob
    from nsr_hip import check, lib, ptr, stream_ptr
    net = _net(35, 13, 1, True, True, seed=3)
    vb = VanillaBlob(_linears(net), 35, 13, activation=1)
    blob = vb.build()
    g = torch.Generator(device="cuda").manual_seed(5)
    nt = n * (1 + n_taps)
    x01 = torch.rand(nt, 3, device="cuda", generator=g)
    enc = (torch.randn(nt, 32, device="cuda", generator=g) * 0.1).half()
    d_out = torch.randn(n, 13, device="cuda", generator=g)
    d_col0 = torch.randn(nt - n, device="cuda", generator=g)
    P = torch.randn(nt, 36, device="cuda", generator=g) * 0.2
    P[:, 35] = 0
    inp = torch.cat([x01 * 2 - 1, enc.float()], -1).requires_grad_(True)
    want = net(inp)
    (gin,) = torch.autograd.grad(want[:, 0].sum(), inp, create_graph=True)
    second = n_taps == 0
    loss = (want[:n] * d_out).sum() + (want[n:, 0] * d_col0).sum()
    if second:
        loss = loss + (gin * P[:, :35]).sum()
    loss.backward()
    d, s = vb.desc, stream_ptr()
    out = torch.empty(n, 16, device="cuda")
    col0 = torch.empty(max(nt - n, 1), device="cuda")
    g_in = torch.empty(nt, 36, device="cuda")
    check(lib.nsr_vmlp_forward(ctypes.byref(d), ptr(blob.detach()), ptr(x01), 3, ptr(enc), 32, ptr(out), ptr(col0),
                               ptr(g_in) if second else None, nt, n, None, s), "fwd")
    assert torch.allclose(out[:, :13], want[:n].detach(), rtol=1e-5, atol=1e-5)
    if n_taps:
        assert torch.allclose(col0, want[n:, 0].detach(), rtol=1e-5, atol=1e-5)
    if second:
        assert _rel(g_in[:, :35], gin.detach()) < 1e-5, _rel(g_in[:, :35], gin.detach())
    d16 = torch.zeros(n, 16, device="cuda")
    d16[:, :13] = d_out
    d_enc = torch.empty(16 * nt * 2, device="cuda")
    gb = torch.empty(vb.n_floats, device="cuda")
    ws = torch.empty(int(lib.nsr_vmlp_backward_workspace_floats(ctypes.byref(d), nt)), device="cuda")
    check(lib.nsr_vmlp_backward(ctypes.byref(d), ptr(blob.detach()), ptr(x01), 3, ptr(enc), 32, ptr(d16), ptr(d_col0),
                                ptr(P) if second else None, ptr(d_enc), 0, 3, 32, 2, ptr(gb), 0, ptr(ws), nt, n, None, s),
          "bwd")
    # level-major [16][nt][2] -> row-major columns 3..34 of d loss / d input
    got = d_enc.view(16, nt, 2).permute(1, 0, 2).reshape(nt, 32)
    assert _rel(got, inp.grad[:, 3:]) < 2e-4, _rel(got, inp.grad[:, 3:])
    want_grads = [p.grad.clone() for p in net.parameters()]
    net.zero_grad()
    vb.push_gradient(gb)
    for (k, p), w in zip(net.named_parameters(), want_grads):
        assert _rel(p.grad, w) < 3e-4, (k, _rel(p.grad, w), second)


def test_level_major_encoding_input_gives_the_same_output():
    """enc_stride = 0x80000000 | F: the SDF-input loader reads the level-major encoding [L][n][F] of the fused encoders"""
    from nsr.fused_neus import VanillaBlob
    from nsr_hip import check, lib, ptr, stream_ptr
    net = _net(35, 13, 1, True, True, seed=9)
    vb = VanillaBlob(_linears(net), 35, 13, activation=1)
    blob = vb.build(requires_grad=False)
    n = 1234
    x01 = torch.rand(n, 3, device="cuda")
    enc = (torch.randn(n, 32, device="cuda") * 0.1).half()
    enc_lm = enc.view(n, 16, 2).permute(1, 0, 2).contiguous()
    outs = []
    rows = (n + 15) // 16 * 16  # tile-major [rows / 16][L][16][F]: enc_stride = 0x40000000 | F
    enc_tm = torch.zeros(rows, 32, dtype=torch.float16, device="cuda")
    enc_tm[:n] = enc
    enc_tm = enc_tm.view(rows // 16, 16, 16, 2).permute(0, 2, 1, 3).contiguous()
    outs, grads = [], []
    d_out = torch.randn(n, 16, device="cuda")
    ws = torch.empty(int(lib.nsr_vmlp_backward_workspace_floats(ctypes.byref(vb.desc), n)), device="cuda")
    for e, stride in ((enc, 32), (enc_lm, 0x80000000 | 2), (enc_tm, 0x40000000 | 2)):
        out = torch.empty(n, 16, device="cuda")
        check(lib.nsr_vmlp_forward(ctypes.byref(vb.desc), ptr(blob), ptr(x01), 3, ptr(e), stride, ptr(out), None, None, n, n,
                                   None, stream_ptr()), "fwd")
        outs.append(out)
        d_enc, g_blob = torch.empty(32 * n, device="cuda"), torch.empty_like(blob)
        check(lib.nsr_vmlp_backward(ctypes.byref(vb.desc), ptr(blob), ptr(x01), 3, ptr(e), stride, ptr(d_out), None, None,
                                    ptr(d_enc), 0, 3, 32, 2, ptr(g_blob), 0, ptr(ws), n, n, None, stream_ptr()), "bwd")
        grads.append((d_enc, g_blob))
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    for d_enc, g_blob in grads[1:]:  # the backward reads the same inputs through the same loader
        assert torch.equal(d_enc, grads[0][0])
        assert torch.allclose(g_blob, grads[0][1], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("n_in,n_out,n_hidden,weight_norm", [(35, 13, 1, True), (32, 3, 2, False), (11, 13, 1, True),
                                                            (24, 3, 2, False), (32, 8, 1, False)])
def test_blob_fold_and_gradient_unfold_match_torch(n_in, n_out, n_hidden, weight_norm):
    """nsr_vmlp_fold / nsr_vmlp_unfold_gradient (one launch each) == the torch formulation of the same arithmetic:
    W = g v / |v| per row (old-style weight_norm, models/network_utils.py:133-137), padded blob, and its autograd"""
    import torch.nn as nn
    from nsr.fused_neus import VanillaBlob, _linear_weight
    torch.manual_seed(0)
    dims = [n_in] + [64] * n_hidden + [n_out]
    layers = []
    for i in range(len(dims) - 1):
        lin = nn.Linear(dims[i], dims[i + 1]).cuda()
        with torch.no_grad():
            lin.bias.normal_(0, 0.3)
        layers.append(nn.utils.weight_norm(lin) if weight_norm else lin)
    if weight_norm:
        with torch.no_grad():
            for lin in layers:
                lin.weight_g.mul_(1.7)
    vb = VanillaBlob(layers, n_in, n_out, activation=0)
    blob = vb.build().clone()
    d = vb.desc
    # torch formulation of the blob
    parts = []
    w0 = _linear_weight(layers[0]).float()
    parts += [torch.nn.functional.pad(w0, (0, d.in_pad - w0.shape[1])).reshape(-1), layers[0].bias]
    for lin in layers[1:-1]:
        parts += [_linear_weight(lin).float().reshape(-1), lin.bias]
    wl = _linear_weight(layers[-1]).float()
    parts += [torch.nn.functional.pad(wl, (0, 0, 0, 16 - wl.shape[0])).reshape(-1),
              torch.nn.functional.pad(layers[-1].bias, (0, 16 - wl.shape[0]))]
    ref = torch.cat(parts)
    assert ref.numel() == blob.numel()
    assert torch.allclose(blob, ref.detach(), rtol=2e-6, atol=1e-7), float((blob - ref.detach()).abs().max())
    gb = torch.randn_like(ref)
    ref.backward(gb)
    names = [(lin, k) for lin in layers for k, _ in lin.named_parameters()]
    want = [dict(lin.named_parameters())[k].grad.clone() for lin, k in names]
    for lin in layers:
        for p in lin.parameters():
            p.grad = None
    vb.push_gradient(gb)   # fresh gradients: written
    vb.push_gradient(gb)   # existing gradients: accumulated
    for (lin, k), ref_g in zip(names, want):
        g = dict(lin.named_parameters())[k].grad
        assert torch.allclose(g, 2 * ref_g, rtol=2e-5, atol=2e-6), (k, float((g - 2 * ref_g).abs().max()))
