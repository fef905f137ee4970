"""fp32 MLP kernels (csrc/vmlp.hip) in isolation on the shapes of the NeuS steps, N samples (default 262,144): the SDF network
over 7 N points (C5: finite-difference taps), over N points with / without second-order terms (C3 / C4), the two-hidden-layer
colour head (C4 / C5).  HIP events, median of 50; gradients checked against fp32 autograd of the same nn.Linear stack.
One JSON line.  NSR_HIP_LIB selects another build of the library (A/B of compile-time switches)."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import torch
from nsr.fused_neus import VanillaBlob
from nsr_hip import check, lib, ptr, stream_ptr
from kernel_microbench import median_us
from test_gpu_vmlp import _net, _linears, _rel

N = int(os.environ.get("N", 262144))
CHECK = not os.environ.get("NO_CHECK")
res = {"N": N, "lib": os.environ.get("NSR_HIP_LIB", "default")}
net = _net(35, 13, 1, True, True, seed=9)
vb = VanillaBlob(_linears(net), 35, 13, activation=1)
blob = vb.build(requires_grad=False)
d = vb.desc
for tag, taps, second in (("sdf_taps_7N", 6, False), ("sdf_second_N", 0, True), ("sdf_plain_N", 0, False)):
    n = N * (1 + taps)
    x01 = torch.rand(n, 3, device="cuda")
    enc = (torch.randn(n, 32, device="cuda") * 0.1).half()
    tm = enc.view(n // 16, 16, 16, 2).permute(0, 2, 1, 3).contiguous()
    d_out = torch.randn(N, 16, device="cuda"); d_out[:, 13:] = 0
    d_col0 = torch.randn(max(n - N, 1), device="cuda")
    P = torch.randn(n, 36, device="cuda") * 0.2; P[:, 35] = 0
    ws = torch.empty(int(lib.nsr_vmlp_backward_workspace_floats(ctypes.byref(d), n)), device="cuda")
    d_enc, gb = torch.empty(32 * n, device="cuda"), torch.empty_like(blob)
    out = torch.empty(N, 16, device="cuda"); col0 = torch.empty(max(n - N, 1), device="cuda")
    g_in = torch.empty(n, 36, device="cuda") if second else None
    b = lambda: check(lib.nsr_vmlp_backward(ctypes.byref(d), ptr(blob), ptr(x01), 3, ptr(tm), 0x40000000 | 2, ptr(d_out),
                                            ptr(d_col0), ptr(P) if second else None, ptr(d_enc), 0, 3, 32, 2, ptr(gb), 0,
                                            ptr(ws), n, N, None, stream_ptr()), "bwd")
    f = lambda: check(lib.nsr_vmlp_forward(ctypes.byref(d), ptr(blob), ptr(x01), 3, ptr(tm), 0x40000000 | 2, ptr(out),
                                           ptr(col0), ptr(g_in) if second else None, n, N, None, stream_ptr()), "fwd")
    res[tag] = {"backward_us": round(median_us(b), 1), "forward_us": round(median_us(f), 1)}
    if CHECK:
        b(); torch.cuda.synchronize()
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
        errs["params_max"] = max(_rel(p.grad, w) for p, w in zip(net.parameters(), want_grads))
        net.zero_grad()
        res[tag]["rel_err_vs_autograd"] = {k: float(f"{v:.2e}") for k, v in errs.items()}
        del inp, want, loss, got
net2 = _net(32, 3, 2, False, False, seed=4)
vb2 = VanillaBlob(_linears(net2), 32, 3, activation=0)
blob2 = vb2.build(requires_grad=False)
d2 = vb2.desc
x = torch.randn(N, 32, device="cuda")
d_out = torch.zeros(N, 16, device="cuda"); d_out[:, :3] = torch.randn(N, 3, device="cuda")
ws = torch.empty(int(lib.nsr_vmlp_backward_workspace_floats(ctypes.byref(d2), N)), device="cuda")
d_x, gb2 = torch.empty(N, 32, device="cuda"), torch.empty_like(blob2)
out2 = torch.empty(N, 16, device="cuda")
b = lambda: check(lib.nsr_vmlp_backward(ctypes.byref(d2), ptr(blob2), ptr(x), 32, None, 0, ptr(d_out), None, None, ptr(d_x),
                                        32, 0, 32, 0, ptr(gb2), 0, ptr(ws), N, N, None, stream_ptr()), "bwd")
f = lambda: check(lib.nsr_vmlp_forward(ctypes.byref(d2), ptr(blob2), ptr(x), 32, None, 0, ptr(out2), None, None, N, N, None,
                                       stream_ptr()), "fwd")
res["colour_N"] = {"backward_us": round(median_us(b), 1), "forward_us": round(median_us(f), 1)}
print(json.dumps(res))
