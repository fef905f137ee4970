"""Duration of the table backward's accumulation launched over single levels / level ranges (items binned once), on
positions and gradients captured from training steps (tools/dump_step_inputs.py): which levels the launch waits for.

    NSR_VARIANT_DATA=build/step_inputs.pt python tools/table_backward_ranges.py
"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd"), os.path.join(ROOT, "tools")]
import torch
import nsr_hip
from nsr_hip import check, lib, ptr, stream_ptr
from kernel_microbench import median_us

if __name__ == "__main__":
    gd = nsr_hip.make_grid_desc(16, 2, 19, 16, 1.447269237440378)
    D = ctypes.byref(gd)
    real = torch.load(os.environ["NSR_VARIANT_DATA"])
    for kv in filter(None, os.environ.get("NSR_OWN_TUNE", "").split(",")):
        pass
    out = {}
    for name, d in real.items():
        x, dy = d["x"].cuda().contiguous(), d["dy"].cuda().contiguous()
        n = x.shape[0]
        g = torch.empty(gd.n_entries * 2, device="cuda")
        ws = torch.empty(int(lib.nsr_hashgrid_backward_params_workspace_floats(D, n)), device="cuda")
        check(lib.nsr_hashgrid_backward_params_owner_bin(ptr(x), ptr(ws), n, 16, D, None, stream_ptr()), "bin")
        res = {}
        for lo, hi in [(l, l + 1) for l in range(16)] + [(0, 4), (4, 16), (5, 16), (0, 16)]:
            def f():
                check(lib.nsr_hashgrid_backward_params_owner_accumulate_range(ptr(x), ptr(dy), ptr(g), None, ptr(ws), n, 16, 1.0,
                                                                              lo, hi, D, None, stream_ptr()), "range")
            res[f"{lo}-{hi}"] = round(median_us(f, 3, 15), 1)
        out[f"{name}:{n}"] = res
    print(json.dumps(out))
