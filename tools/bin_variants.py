"""A/B of the binning pass of the table backward (k_own_bin) across builds (tools/build_variants.sh): time of the binning
launch alone and of binning + accumulation, synthetic ray-coherent positions at the bench's two operating points and the
positions captured from training steps (NSR_VARIANT_DATA); the accumulated gradient's fp64 sum so that builds can be
compared.  One JSON line per build.

    python tools/bin_variants.py build/variants/libnsr_hip_b*.so
"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker():
    sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd"), os.path.join(ROOT, "tools")]
    import ctypes
    import torch
    import nsr_hip
    from nsr_hip import check, lib, ptr, stream_ptr
    from kernel_microbench import coherent, median_us
    gd = nsr_hip.make_grid_desc(16, 2, 19, 16, 1.447269237440378)
    cases = {}
    for n in (96000, 216000, 1000000):
        cases[str(n)] = (coherent((n + 63) // 64 * 64, per_ray=16)[:n].contiguous(),
                         torch.randn(16, n, 2, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)) * 1e-3)
    if os.environ.get("NSR_VARIANT_DATA"):
        for k, v in torch.load(os.environ["NSR_VARIANT_DATA"]).items():
            cases[f"real_{k}_{v['x'].shape[0]}"] = (v["x"].cuda().contiguous(), v["dy"].cuda().contiguous())
    res = {"lib": os.path.basename(nsr_hip.LIB_PATH)}
    for name, (x, dy) in cases.items():
        n = x.shape[0]
        g = torch.empty(gd.n_entries * 2, device="cuda")
        ws = torch.empty(int(lib.nsr_hashgrid_backward_params_workspace_floats(ctypes.byref(gd), n)), device="cuda")

        def bin_():
            check(lib.nsr_hashgrid_backward_params_owner_bin(ptr(x), ptr(ws), n, 16, ctypes.byref(gd), None, stream_ptr()), "bin")

        def acc():
            check(lib.nsr_hashgrid_backward_params_owner_accumulate(ptr(x), ptr(dy), 2, 0, ptr(g), ptr(ws), n, 16, 1.0, 0,
                                                                    ctypes.byref(gd), None, stream_ptr()), "acc")

        def both():
            bin_(); acc()
        both()
        res[name] = {"bin_us": round(median_us(bin_, 10, 40), 1), "accumulate_us": round(median_us(acc, 10, 40), 1),
                     "both_us": round(median_us(both, 10, 40), 1), "grad_sum": float(g.double().sum()),
                     "grad_abs_sum": float(g.double().abs().sum())}
    print("RESULT " + json.dumps(res))


if __name__ == "__main__":
    if os.environ.get("NSR_BIN_VARIANT_WORKER"):
        worker()
    else:
        for libp in sys.argv[1:]:
            p = subprocess.run([sys.executable, os.path.abspath(__file__)], capture_output=True, text=True,
                               env=dict(os.environ, NSR_HIP_LIB=os.path.abspath(libp), NSR_BIN_VARIANT_WORKER="1"))
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
            print(line[-1][7:] if line else json.dumps({"lib": libp, "error": (p.stderr or p.stdout)[-400:]}), flush=True)
