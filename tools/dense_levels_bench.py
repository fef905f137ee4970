"""Isolated timings of the table backward's pieces on CAPTURED training-step inputs (build/step_inputs.pt, tools/dump_step_inputs.py):
owner workgroups over all levels vs the dense-level path (csrc/hashgrid_dense.inc) + the owner over the hashed levels only,
each with and without AdamW inside; the number of cell runs (= atomics / 16) the ray-run merge leaves per level.
    python tools/dense_levels_bench.py   (NSR_DENSE_PROBE=1: the dense kernel without its atomics, 2: fp32 atomics)"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd"), os.path.join(ROOT, "tools")]
import torch
import nsr_hip
from nsr_hip import NsrTableAdam, check, lib, ptr, stream_ptr
from kernel_microbench import median_us

gd = nsr_hip.make_grid_desc(16, 2, 19, 16, 1.447269237440378)
D = int(lib.nsr_hashgrid_dense_levels(ctypes.byref(gd)))
real = torch.load(os.environ.get("NSR_VARIANT_DATA", os.path.join(ROOT, "build", "step_inputs.pt")))
n_tab = gd.n_entries * 2
res = {"dense_levels": D, "probe": os.environ.get("NSR_DENSE_PROBE", "0"), "cases": {}}
p = torch.randn(n_tab, device="cuda") * 0.1
st = dict(p=p, m=torch.zeros_like(p), v=torch.zeros_like(p), h=torch.empty(n_tab, dtype=torch.float16, device="cuda"),
          step=torch.zeros(1, dtype=torch.int32, device="cuda"), hyper=torch.zeros(12, device="cuda"))
ad = NsrTableAdam()
ad.params, ad.exp_avg, ad.exp_avg_sq, ad.shadow = st["p"].data_ptr(), st["m"].data_ptr(), st["v"].data_ptr(), st["h"].data_ptr()
ad.step, ad.hyper = st["step"].data_ptr(), st["hyper"].data_ptr()
ad.base_lr, ad.beta1, ad.beta2, ad.gamma = 0.01, 0.9, 0.99, 0.33
ad.milestone0, ad.milestone1, ad.milestone2 = 10000, 15000, 18000
ad.eps, ad.weight_decay = 1e-15, 0.01
for name, d in real.items():
    x, dy = d["x"].cuda().contiguous(), d["dy"].cuda().contiguous()
    n = x.shape[0]
    g = torch.empty(n_tab, device="cuda")
    ws = torch.empty(int(lib.nsr_hashgrid_backward_params_workspace_floats(ctypes.byref(gd), n)), device="cuda")
    s = stream_ptr()
    B, G = ctypes.byref(gd), ctypes.byref(ad)
    f = {
        "bin_all": lambda: check(lib.nsr_hashgrid_backward_params_owner_bin(ptr(x), ptr(ws), n, 16, B, None, s), "bin"),
        "owner_all_grad": lambda: check(lib.nsr_hashgrid_backward_params_owner_accumulate(ptr(x), ptr(dy), 2, 0, ptr(g), ptr(ws), n, 16, 1.0, 0, B, None, s), "a"),
        "owner_all_adam": lambda: check(lib.nsr_hashgrid_backward_params_owner_accumulate_adam(ptr(x), ptr(dy), 2, 0, ptr(ws), n, 16, 1.0, B, None, G, s), "a"),
        "bin_hashed": lambda: check(lib.nsr_hashgrid_backward_params_owner_bin_range(ptr(x), ptr(ws), n, 16, D, 16, B, None, s), "bin"),
        "owner_hashed_grad": lambda: check(lib.nsr_hashgrid_backward_params_owner_accumulate_range(ptr(x), ptr(dy), ptr(g), None, ptr(ws), n, 16, 1.0, D, 16, B, None, s), "a"),
        "owner_hashed_adam": lambda: check(lib.nsr_hashgrid_backward_params_owner_accumulate_adam_range(ptr(x), ptr(dy), ptr(ws), n, 16, 1.0, D, 16, B, None, G, s), "a"),
        "dense_clear": lambda: check(lib.nsr_hashgrid_backward_params_dense(None, None, None, None, None, ptr(ws), n, 16, 1.0, 0, B, None, 1, s), "d"),
        "dense_accumulate": lambda: check(lib.nsr_hashgrid_backward_params_dense(ptr(x), ptr(dy), None, None, None, ptr(ws), n, 16, 1.0, 0, B, None, 2, s), "d"),
        "dense_finish_grad": lambda: check(lib.nsr_hashgrid_backward_params_dense(None, None, ptr(g), None, None, ptr(ws), n, 16, 1.0, 0, B, None, 4, s), "d"),
        "dense_finish_adam": lambda: check(lib.nsr_hashgrid_backward_params_dense(None, None, None, None, G, ptr(ws), n, 16, 1.0, 0, B, None, 4, s), "d"),
    }
    out = {"n": n}
    f["bin_all"]()
    for k in ("bin_all", "owner_all_grad", "owner_all_adam"):
        out[k + "_us"] = round(median_us(f[k], 5, 20), 1)
    f["bin_hashed"](); f["dense_clear"]()
    for k in ("bin_hashed", "owner_hashed_grad", "owner_hashed_adam", "dense_clear", "dense_accumulate", "dense_finish_grad",
              "dense_finish_adam"):
        out[k + "_us"] = round(median_us(f[k], 5, 20), 1)
    # cell runs per level along the sample order, as the kernel forms them (64-lane waves, blocks of 256)
    runs = {}
    for lvl in range(D):
        sc, r = float(gd.scale[lvl]), int(gd.resolution[lvl])
        c = torch.floor(x * sc + 0.5).long()
        key = c[:, 0] + r * (c[:, 1] + r * c[:, 2])
        head = torch.ones(n, dtype=torch.bool, device="cuda")
        head[1:] = key[1:] != key[:-1]
        head[::64] = True
        runs[lvl] = int(head.sum())
    out["runs_per_level"] = runs
    out["atomics"] = 16 * sum(runs.values())
    res["cases"][name] = out
print(json.dumps(res))
