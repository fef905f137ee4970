"""fp32 SDF network (35 -> 64 -> 13) forward / backward in isolation with the encoding in the three layouts the loader
accepts: row-major, level-major, tile-major.  One JSON line."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")]
import torch
from nsr.fused_neus import VanillaBlob
from nsr_hip import check, lib, ptr, stream_ptr
from kernel_microbench import median_us
from test_gpu_vmlp import _net, _linears

net = _net(35, 13, 1, True, True, seed=9)
vb = VanillaBlob(_linears(net), 35, 13, activation=1)
blob = vb.build(requires_grad=False)
res = {}
for n in (262144, 1048576):
    x01 = torch.rand(n, 3, device="cuda")
    enc = (torch.randn(n, 32, device="cuda") * 0.1).half()
    lm = enc.view(n, 16, 2).permute(1, 0, 2).contiguous()
    tm = enc.view(n // 16, 16, 16, 2).permute(0, 2, 1, 3).contiguous()
    d_out = torch.randn(n, 16, device="cuda")
    ws = torch.empty(int(lib.nsr_vmlp_backward_workspace_floats(ctypes.byref(vb.desc), n)), device="cuda")
    out = torch.empty(n, 16, device="cuda")
    d_enc, g_blob = torch.empty(32 * n, device="cuda"), torch.empty_like(blob)
    for name, e, stride in (("row", enc, 32), ("level", lm, 0x80000000 | 2), ("tile", tm, 0x40000000 | 2)):
        f = lambda: check(lib.nsr_vmlp_forward(ctypes.byref(vb.desc), ptr(blob), ptr(x01), 3, ptr(e), stride, ptr(out), None,
                                               None, n, n, None, stream_ptr()), "fwd")
        b = lambda: check(lib.nsr_vmlp_backward(ctypes.byref(vb.desc), ptr(blob), ptr(x01), 3, ptr(e), stride, ptr(d_out), None,
                                                None, ptr(d_enc), 0, 3, 32, 2, ptr(g_blob), 0, ptr(ws), n, n, None,
                                                stream_ptr()), "bwd")
        res[f"{name}:{n}"] = {"forward_us": round(median_us(f), 1), "backward_us": round(median_us(b), 1)}
print(json.dumps(res))
