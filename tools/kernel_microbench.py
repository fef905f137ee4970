"""Kernel-level micro-benchmarks of SURVEY.md section 8(d): hash-grid encode forward / forward+backward(table) / fused
encode+MLP on E1 (uniform) and E2 (ray-coherent) inputs at N in {2^18, 2^20, 2^22}, C2 shapes (L16 T2^19 F2, 64-wide MLP),
plus the two measured ceilings the roofline fractions refer to: a device stream copy (GB/s) and a dense fp16 GEMM on the
MFMA units (TFLOP/s, hipBLASLt through torch).  20 warm-up + 50 timed launches, HIP events, median.  One JSON to stdout.

    python tools/kernel_microbench.py > profiles/rNN_microbench.json
"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd")]
import torch
import nsr_hip
from nsr_hip import check, lib, ops, ptr, stream_ptr


def median_us(fn, warm=20, iters=50):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def coherent(n, per_ray=64, step=0.00507421875 / 3.0):
    r = n // per_ray
    g = torch.Generator(device="cuda").manual_seed(0)
    o = torch.rand(r, 1, 3, device="cuda", generator=g) * 0.6 + 0.2
    d = torch.nn.functional.normalize(torch.randn(r, 1, 3, device="cuda", generator=g), dim=-1)
    t = torch.arange(per_ray, device="cuda").view(1, -1, 1) * step
    return (o + d * t).clamp(0, 1).reshape(-1, 3).contiguous()


def main():
    gd = nsr_hip.make_grid_desc(16, 2, 19, 16, 1.447269237440378)
    md = nsr_hip.make_mlp_desc(32, 16, 1, "none")
    torch.manual_seed(1337)
    table = (torch.randn(gd.n_entries * 2, device="cuda") * 0.1).half()
    w = (torch.randn(64 * 32 + 1024, device="cuda") * 0.1).half()
    res = {"device": torch.cuda.get_device_name(0), "shapes": "HashGrid L16 T2^19 F2 (nerf-blender) + 32->64->16 fused MLP",
           "bytes_per_sample": {"encode_fwd": 588, "encode_bwd_table": 2124}, "cases": []}
    # ceilings
    n_copy = 1 << 28
    src, dst = torch.empty(n_copy, dtype=torch.float32, device="cuda"), torch.empty(n_copy, dtype=torch.float32, device="cuda")
    t = median_us(lambda: dst.copy_(src), 5, 20)
    res["stream_copy_GBps"] = 2 * 4 * n_copy / t / 1e3
    a = torch.randn(8192, 8192, device="cuda", dtype=torch.float16)
    b = torch.randn(8192, 8192, device="cuda", dtype=torch.float16)
    t = median_us(lambda: torch.matmul(a, b), 5, 20)
    res["mfma_fp16_gemm_TFLOPs"] = 2 * 8192 ** 3 / t / 1e6
    del src, dst, a, b
    for logn in (18, 20, 22):
        n = 1 << logn
        for dist in ("E1_uniform", "E2_coherent"):
            x = torch.rand(n, 3, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0)) if dist == "E1_uniform" else coherent(n)
            y = torch.empty(n, 32, dtype=torch.float16, device="cuda")
            t_fwd = median_us(lambda: ops.hashgrid_forward(x, table, gd, out=y))
            dy = torch.randn(16, n, 2, device="cuda")
            g = torch.empty(gd.n_entries * 2, device="cuda")
            nws = int(lib.nsr_hashgrid_backward_params_workspace_floats(ctypes.byref(gd), n))
            ws = torch.empty(nws, device="cuda")

            def bwd():
                check(lib.nsr_hashgrid_backward_params_owner(ptr(x), ptr(dy), 2, 0, ptr(g), ptr(ws), n, 16, 1.0, 0,
                                                             ctypes.byref(gd), None, stream_ptr()), "owner")
            t_bwd = median_us(bwd)
            out = torch.empty(n, 16, dtype=torch.float16, device="cuda")

            def fused():
                ops.hashgrid_forward(x, table, gd, out=y)
                check(lib.nsr_mlp_forward(ptr(y), 0, 32, ptr(w), ptr(out), None, n, ctypes.byref(md), stream_ptr()), "mlp")
            t_fused = median_us(fused)
            res["cases"].append({"n": n, "inputs": dist, "encode_fwd_us": t_fwd, "encode_bwd_table_us": t_bwd,
                                 "encode_plus_mlp_fwd_us": t_fused,
                                 "encode_fwd_GBps_algorithmic": 588 * n / t_fwd / 1e3,
                                 "encode_bwd_GBps_algorithmic": 2124 * n / t_bwd / 1e3,
                                 "encode_fwd_samples_per_s": n / t_fwd * 1e6})
            del x, y, dy, ws, out
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
