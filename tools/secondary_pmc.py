"""Workload for the PMC passes behind bench.py's `roofline_secondary`: what BASELINE.json's north_star asks to be evidenced by
rocprof -- HBM GB/s on the hash gather and MFMA utilisation on the fused fp16 MLP.  2^18 samples, the nerf-blender density
network (HashGrid L16 T2^19 F2 -> 32 -> 64 -> 16): 20 launches each of the stand-alone encode (k_grid_forward_pair), the
fused MLP forward, its backward (k_mlp_dgrad + the k_mlp_wgrad kernels) and the one-kernel encode -> MLP, on E1 (uniform) or
E2 (ray-coherent) positions (argv[1]).  Prints the HIP-event medians of the same launches as JSON (argv[2] = out file).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/ps -o s -- python tools/secondary_pmc.py E2
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd"), os.path.join(ROOT, "tools")]
import torch
import nsr_hip
from nsr_hip import ops
from kernel_microbench import coherent, median_us

if __name__ == "__main__":
    kind = sys.argv[1] if len(sys.argv) > 1 else "E2"
    n = 1 << 18
    gd = nsr_hip.make_grid_desc(16, 2, 19, 16, 1.447269237440378)
    md = nsr_hip.make_mlp_desc(32, 16, 1, "none")
    g = torch.Generator().manual_seed(0)
    table = ((torch.rand(gd.n_entries * 2, generator=g) * 2 - 1) * 0.1).half().cuda()
    w = (torch.randn(64 * 32 + 1024, generator=g) * 0.1).half().cuda()
    x = coherent(n, per_ray=64) if kind == "E2" else torch.rand(n, 3, device="cuda")
    dout = torch.randn(n, 16, device="cuda")
    gw = torch.zeros(64 * 32 + 1024, device="cuda")
    import ctypes
    from nsr_hip import check, lib, ptr, stream_ptr
    enc = (torch.randn(n, 32, device="cuda") * 0.1).half()  # (not produced by the encode: its launches are what the counters average)
    enc_lm = torch.empty(16 * n * 2, dtype=torch.float16, device="cuda")
    out, acts = ops.mlp_forward(enc, w, md, save_acts=True)

    def encode_level_major():  # the layout the training step's encode writes ([L][n][F]: csrc/step.hip nsr_nerf_prune_pass)
        check(lib.nsr_hashgrid_forward_ex(ptr(x), ptr(table), ptr(enc_lm), n, 32, 1, 16, ctypes.byref(gd), None, stream_ptr()),
              "nsr_hashgrid_forward_ex")

    res = {"kind": kind, "n": n,
           "hashgrid_forward_us": median_us(encode_level_major, 5, 20),
           "mlp_forward_us": median_us(lambda: ops.mlp_forward(enc, w, md, save_acts=True), 5, 20),
           "mlp_backward_us": median_us(lambda: ops.mlp_backward(dout, out, enc, acts, w, md, grad_weights=gw, want_dx=True,
                                                                 grad_scale=128.0), 5, 20),
           "grid_mlp_forward_us": median_us(lambda: ops.grid_mlp_forward(x, table, w, gd, md), 5, 20)}
    # round 6: the step's own backward kernels -- both networks' data gradients in one launch (k_mlp_dgrad_pair) and the
    # weight-gradient kernels behind it, on the nerf-blender shapes (density 32 -> 64 -> 16, colour 32 -> 64 -> 64 -> 3)
    dc, dd = nsr_hip.make_mlp_desc(32, 3, 2, "sigmoid"), md
    wc = (torch.randn(64 * 32 + 4096 + 1024, generator=g) * 0.1).half().cuda()
    enc_l = torch.randn(16, n, 2, generator=g).half().cuda()
    o1, a1 = torch.empty(n, 16).half().cuda(), torch.empty(1, n, 64).half().cuda()
    s_ = stream_ptr()
    check(lib.nsr_mlp_forward_ex(ptr(enc_l), 0, 32, 2, ptr(w), ptr(o1), ptr(a1), n, ctypes.byref(dd), None, s_), "fwd")
    tex = torch.cat([o1, torch.rand(n, 16, generator=g).half().cuda()], 1).contiguous()
    o2, a2 = ops.mlp_forward(tex, wc, dc, save_acts=True)
    dr, dl = (torch.randn(n, 3, generator=g) * 1e-3).cuda(), (torch.randn(n, generator=g) * 1e-3).cuda()
    wsz = lambda d: torch.zeros(int(lib.nsr_mlp_backward_workspace_floats(ctypes.byref(d), n)), device="cuda")  # noqa: E731
    pc, pd = wsz(dc), wsz(dd)
    gc, gdd = torch.zeros_like(wc, dtype=torch.float32), torch.zeros_like(w, dtype=torch.float32)
    denc = torch.zeros(16, n, 2).cuda()

    def pair():
        check(lib.nsr_mlp_dgrad_pair(ptr(dr), ptr(dl), ptr(o2), ptr(a2), ptr(wc), ptr(pc), ptr(a1), ptr(w), ptr(pd), ptr(denc),
                                     n, 65536.0, ctypes.byref(dc), ctypes.byref(dd), None, s_), "pair")

    def wgrads():
        check(lib.nsr_mlp_backward_phases(ptr(dr), 1, 3, None, ptr(o2), ptr(tex), 0, 32, 0, ptr(a2), ptr(wc), ptr(gc), None, 32, 0,
                                          ptr(pc), n, 65536.0, ctypes.byref(dc), None, s_, 2), "c")
        check(lib.nsr_mlp_backward_phases(ptr(denc), 1, 32, ptr(dl), ptr(o1), ptr(enc_l), 0, 32, 2, ptr(a1), ptr(w), ptr(gdd), None,
                                          32, 2, ptr(pd), n, 65536.0, ctypes.byref(dd), None, s_, 2), "d")

    res["dgrad_pair_us"] = median_us(pair, 5, 20)
    res["wgrad_both_networks_us"] = median_us(wgrads, 5, 20)
    torch.cuda.synchronize()
    if len(sys.argv) > 2:
        json.dump(res, open(sys.argv[2], "w"))
    print(json.dumps(res))
