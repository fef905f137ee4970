"""fp32 MLP backward (csrc/vmlp.hip) in isolation on the shapes of the NeuS steps: the SDF network over 7 N points (C5:
finite-difference taps), over N points with second-order terms (C3 / C4), the two-hidden-layer colour head (C4 / C5).
Times the split form (k_vmlp_dgrad + k_vmlp_wgrad) for several wave counts, the forward kernel beside it, and checks the
gradients against fp32 autograd of the same nn.Linear stack.  One JSON line."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import torch
from nsr.fused_neus import VanillaBlob
from nsr_hip import check, lib, ptr, stream_ptr
from kernel_microbench import median_us
from test_gpu_vmlp import _net, _linears, _rel

N = int(os.environ.get("N", 262144))
res = {"N": N}


def tile_major(enc):
    n = enc.shape[0]
    return enc.view(n // 16, 16, 16, 2).permute(0, 2, 1, 3).contiguous()


WS = {}


def run(tag, make_fn, desc, n, settings):
    """make_fn(ws) -> the launch; the workspace is sized AFTER the wave counts are set (they are part of its layout)"""
    out = {}
    for name, (split, ww, dw) in settings.items():
        check(lib.nsr_vmlp_tune(0, split), "tune")
        check(lib.nsr_vmlp_tune(1, ww), "tune"); check(lib.nsr_vmlp_tune(2, dw), "tune")
        ws = torch.empty(int(lib.nsr_vmlp_backward_workspace_floats(ctypes.byref(desc), n)), device="cuda")
        out[name] = round(median_us(make_fn(ws)), 1)
        del ws
    check(lib.nsr_vmlp_tune(0, 1), "tune"); check(lib.nsr_vmlp_tune(1, 2048), "t"); check(lib.nsr_vmlp_tune(2, 2048), "t")
    res[tag] = out


SET = {"split_2048": (1, 2048, 2048), "split_1024": (1, 1024, 1024), "split_4096_2048": (1, 4096, 2048),
       "split_2048_3072": (1, 2048, 3072), "fused": (0, 2048, 2048)}
if os.environ.get("ONLY_SPLIT"):
    SET.pop("fused")

# ---- SDF network 35 -> 64 -> 13, softplus, weight norm ----
net = _net(35, 13, 1, True, True, seed=9)
vb = VanillaBlob(_linears(net), 35, 13, activation=1)
blob = vb.build(requires_grad=False)
d = vb.desc
for tag, taps, second in (("sdf_taps_7N", 6, False), ("sdf_second_N", 0, True), ("sdf_plain_N", 0, False)):
    n = N * (1 + taps)
    x01 = torch.rand(n, 3, device="cuda")
    enc = (torch.randn(n, 32, device="cuda") * 0.1).half()
    tm = tile_major(enc)
    d_out = torch.randn(N, 16, device="cuda"); d_out[:, 13:] = 0
    d_col0 = torch.randn(max(n - N, 1), device="cuda")
    P = torch.randn(n, 36, device="cuda") * 0.2; P[:, 35] = 0
    ws = torch.empty(int(lib.nsr_vmlp_backward_workspace_floats(ctypes.byref(d), n)), device="cuda")
    d_enc, gb = torch.empty(32 * n, device="cuda"), torch.empty_like(blob)
    out = torch.empty(N, 16, device="cuda"); col0 = torch.empty(max(n - N, 1), device="cuda")
    g_in = torch.empty(n, 36, device="cuda") if second else None
    mk = lambda w: (lambda: check(lib.nsr_vmlp_backward(ctypes.byref(d), ptr(blob), ptr(x01), 3, ptr(tm), 0x40000000 | 2,
                                                        ptr(d_out), ptr(d_col0), ptr(P) if second else None, ptr(d_enc), 0, 3,
                                                        32, 2, ptr(gb), 0, ptr(w), n, N, None, stream_ptr()), "bwd"))
    b = mk(ws)
    f = lambda: check(lib.nsr_vmlp_forward(ctypes.byref(d), ptr(blob), ptr(x01), 3, ptr(tm), 0x40000000 | 2, ptr(out),
                                           ptr(col0), ptr(g_in) if second else None, n, N, None, stream_ptr()), "fwd")
    run(tag, mk, d, n, SET)
    res[tag]["forward"] = round(median_us(f), 1)
    b(); torch.cuda.synchronize()  # (default form, default wave counts: what `ws` was sized for)
    inp = torch.cat([x01 * 2 - 1, enc.float()], -1).requires_grad_(True)
    want = net(inp)
    loss = (want[:N] * d_out[:, :13]).sum() + ((want[N:, 0] * d_col0).sum() if taps else 0.0)
    if second:
        (gin,) = torch.autograd.grad(want[:, 0].sum(), inp, create_graph=True)
        loss = loss + (gin * P[:, :35]).sum()
    loss.backward()
    got = d_enc.view(16, n, 2).permute(1, 0, 2).reshape(n, 32)
    errs = {"d_enc": _rel(got, inp.grad[:, 3:])}
    want_grads = [p.grad.clone() for p in net.parameters()]
    net.zero_grad(); vb.push_gradient(gb)
    for (k, p), w in zip(net.named_parameters(), want_grads):
        errs[k] = _rel(p.grad, w)
    net.zero_grad()
    res[tag]["rel_err_vs_autograd"] = {k: float(f"{v:.2e}") for k, v in errs.items()}
    del inp, want, loss, got

# ---- colour head 32 -> 64 -> 64 -> 3, ReLU ----
net2 = _net(32, 3, 2, False, False, seed=4)
vb2 = VanillaBlob(_linears(net2), 32, 3, activation=0)
blob2 = vb2.build(requires_grad=False)
d2 = vb2.desc
x = torch.randn(N, 32, device="cuda")
d_out = torch.zeros(N, 16, device="cuda"); d_out[:, :3] = torch.randn(N, 3, device="cuda")
ws = torch.empty(int(lib.nsr_vmlp_backward_workspace_floats(ctypes.byref(d2), N)), device="cuda")
d_x, gb2 = torch.empty(N, 32, device="cuda"), torch.empty_like(blob2)
mk2 = lambda w: (lambda: check(lib.nsr_vmlp_backward(ctypes.byref(d2), ptr(blob2), ptr(x), 32, None, 0, ptr(d_out), None, None,
                                                     ptr(d_x), 32, 0, 32, 0, ptr(gb2), 0, ptr(w), N, N, None, stream_ptr()), "bwd"))
run("colour_N", mk2, d2, N, SET)
mk2(ws)(); torch.cuda.synchronize()
xr = x.clone().requires_grad_(True)
(net2(xr) * d_out[:, :3]).sum().backward()
errs = {"d_x": _rel(d_x, xr.grad)}
want_grads = [p.grad.clone() for p in net2.parameters()]
net2.zero_grad(); vb2.push_gradient(gb2)
for (k, p), w in zip(net2.named_parameters(), want_grads):
    errs[k] = _rel(p.grad, w)
res["colour_N"]["rel_err_vs_autograd"] = {k: float(f"{v:.2e}") for k, v in errs.items()}
print(json.dumps(res))
