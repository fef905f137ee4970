"""Where the owner-computes table backward spends its time at the bench's operating point (~1e5 kept samples, ray-coherent
positions): the binning passes and the accumulation kernel timed separately, then the accumulation with only the first
k levels active (progressive mask) -- successive differences = marginal cost of each level.  One JSON to stdout.

    python tools/table_backward_levels.py [n_samples]
"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd"), os.path.join(ROOT, "tools")]
import torch
import nsr_hip
from nsr_hip import check, lib, ptr, stream_ptr
from kernel_microbench import coherent, median_us


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    gd = nsr_hip.make_grid_desc(16, 2, 19, 16, 1.447269237440378)
    res = {"n": n, "cases": {}, "lib": os.path.basename(nsr_hip.LIB_PATH)}
    if lib.nsr_hashgrid_owner_tune(0, 0.0) < 0:
        pass  # (a baseline build: the knob is a stub)
    real = torch.load(os.environ["NSR_VARIANT_DATA"]) if os.environ.get("NSR_VARIANT_DATA") else None
    for dist in (tuple(real.keys()) if real else ("E2_coherent", "E1_uniform")):
        if real:  # captured from training steps (tools/dump_step_inputs.py)
            x, dy = real[dist]["x"].cuda().contiguous(), real[dist]["dy"].cuda().contiguous()
            n = x.shape[0]
        else:
            x = coherent((n + 63) // 64 * 64, per_ray=16)[:n].contiguous() if dist == "E2_coherent" else \
                torch.rand(n, 3, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0))
            dy = torch.randn(16, n, 2, device="cuda")
        g = torch.empty(gd.n_entries * 2, device="cuda")
        ws = torch.empty(int(lib.nsr_hashgrid_backward_params_workspace_floats(ctypes.byref(gd), n)), device="cuda")
        out = {}
        for k in list(range(1, 17)):
            def bin_():
                check(lib.nsr_hashgrid_backward_params_owner_bin(ptr(x), ptr(ws), n, k, ctypes.byref(gd), None, stream_ptr()), "bin")

            def acc():
                check(lib.nsr_hashgrid_backward_params_owner_accumulate(ptr(x), ptr(dy), 2, 0, ptr(g), ptr(ws), n, k, 1.0, 0,
                                                                        ctypes.byref(gd), None, stream_ptr()), "acc")
            bin_()
            out[k] = {"bin_us": median_us(bin_, 5, 20), "accumulate_us": median_us(acc, 5, 20)}
        res["cases"][dist] = {"levels_active": out,
                              "marginal_accumulate_us": {k: round(out[k]["accumulate_us"] - (out[k - 1]["accumulate_us"] if k > 1 else 0), 1)
                                                         for k in out},
                              "marginal_bin_us": {k: round(out[k]["bin_us"] - (out[k - 1]["bin_us"] if k > 1 else 0), 1) for k in out}}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
