"""A/B of the owner-computes table backward across builds (tools/build_variants.sh, tools/build_baseline.sh) and across the
run-time knobs of the decomposition (nsr_hashgrid_owner_tune): item binning, accumulation with the gradient stored (fp32 /
bf16 in one or two level groups) and with AdamW applied inside, at the two operating points of bench.py (~9.6e4 kept samples
steady state, ~2.2e5 in the short driver run) and at a NeuS-sized launch, ray-coherent positions.  One JSON line per
(build, setting); per-level fp64 sums of the gradient so that builds can be compared bit for bit offline.

    python tools/table_backward_variants.py build/variants/libnsr_hip_*.so instant-nsr-pl_amd/nsr_hip/libnsr_hip.so
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (name, {tune key: value}); keys of nsr_hashgrid_owner_tune: 0 placement, 1 cost_adam, 2 cost_items, 3 rl_max_res, 4 rl_max_q
SETTINGS = [("default", {}), ("dealt", {0: 0}), ("listed", {0: 1}), ("merge_all_dense", {5: 2}), ("merge_off", {5: 0})]
if os.environ.get("NSR_VARIANT_SETTINGS"):
    SETTINGS = [s_ for s_ in SETTINGS if s_[0] in os.environ["NSR_VARIANT_SETTINGS"].split(",")]
DEFAULTS = {0: 2, 1: 1.0, 2: 3.0, 3: 11, 4: 64, 5: 1}


def worker():
    sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd"), os.path.join(ROOT, "tools")]
    import ctypes
    import torch
    import nsr_hip
    from nsr_hip import check, lib, ptr, stream_ptr
    from kernel_microbench import coherent, median_us
    gd = nsr_hip.make_grid_desc(16, 2, 19, 16, 1.447269237440378)
    P = gd.n_entries * 2
    off = [int(o) * 2 for o in gd.offset[:17]]
    tunable = lib.nsr_hashgrid_owner_tune(0, 1.0) >= 0  # (the stub of a baseline build returns -1)
    sizes = tuple(int(v) for v in os.environ.get("NSR_VARIANT_SIZES", "96000,216000,1000000").split(","))
    real = torch.load(os.environ["NSR_VARIANT_DATA"]) if os.environ.get("NSR_VARIANT_DATA") else None
    if real is not None:  # positions / gradients captured from training steps (tools/dump_step_inputs.py)
        sizes = tuple(real.keys())
    for sname, tune in (SETTINGS if tunable else [("build_default", {})]):
        if tunable:
            for k, v in DEFAULTS.items():
                lib.nsr_hashgrid_owner_tune(k, float(tune.get(k, v)))
        res = {"lib": os.path.basename(nsr_hip.LIB_PATH), "setting": sname}
        for cfg_name, thr in (("small_2^11x256", 0xffffffff), ("large_2^13x1024", 0)):
            lib.nsr_hashgrid_owner_large_from(thr)
            for size in sizes:
                if real is not None:
                    x, dy = real[size]["x"].cuda().contiguous(), real[size]["dy"].cuda().contiguous()
                    n, label = x.shape[0], f"{size}{x.shape[0]}"
                    if cfg_name.startswith("large") and n < 200000:
                        continue
                else:
                    n, label = size, str(size)
                    if (cfg_name.startswith("small") and n > 300000) or (cfg_name.startswith("large") and n < 200000):
                        continue  # (each configuration at the sizes it is picked for, and both at the crossover)
                    x = coherent((n + 63) // 64 * 64, per_ray=16)[:n].contiguous()
                    dy = torch.randn(16, n, 2, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)) * 1e-3
                g = torch.empty(P, device="cuda")
                g16 = torch.empty(P + 64, dtype=torch.bfloat16, device="cuda")
                ws = torch.empty(int(lib.nsr_hashgrid_backward_params_workspace_floats(ctypes.byref(gd), n)), device="cuda")
                p, m, v = torch.randn(P, device="cuda") * 1e-2, torch.zeros(P, device="cuda"), torch.zeros(P, device="cuda")
                sh = torch.empty(P, dtype=torch.float16, device="cuda")
                step, hyper = torch.zeros(1, dtype=torch.int32, device="cuda"), torch.zeros(12, device="cuda")
                ad = nsr_hip.NsrTableAdam()
                ad.params, ad.exp_avg, ad.exp_avg_sq, ad.shadow = p.data_ptr(), m.data_ptr(), v.data_ptr(), sh.data_ptr()
                ad.step, ad.hyper = step.data_ptr(), hyper.data_ptr()
                ad.base_lr, ad.beta1, ad.beta2, ad.gamma = 0.01, 0.9, 0.99, 0.33
                ad.milestone0, ad.milestone1, ad.milestone2 = 10000, 15000, 18000
                ad.eps, ad.weight_decay = 1e-15, 0.01
                D = ctypes.byref(gd)

                def bin_():
                    check(lib.nsr_hashgrid_backward_params_owner_bin(ptr(x), ptr(ws), n, 16, D, None, stream_ptr()), "bin")

                def acc():
                    check(lib.nsr_hashgrid_backward_params_owner_accumulate(ptr(x), ptr(dy), 2, 0, ptr(g), ptr(ws), n, 16, 1.0, 0,
                                                                            D, None, stream_ptr()), "acc")

                def acc_adam():
                    check(lib.nsr_hashgrid_backward_params_owner_accumulate_adam(ptr(x), ptr(dy), 2, 0, ptr(ws), n, 16, 1.0, D,
                                                                                 None, ctypes.byref(ad), stream_ptr()), "acc_adam")

                def acc_bf16(groups):
                    def f():
                        for lo, hi in groups:
                            check(lib.nsr_hashgrid_backward_params_owner_accumulate_range(ptr(x), ptr(dy), None, ptr(g16), ptr(ws), n,
                                                                                          16, 1.0, lo, hi, D, None, stream_ptr()),
                                  "range")
                    return f

                def timed(fn, warm=5, iters=30):
                    """median of `fn` alone, the items re-binned (untimed) in front of every launch: an accumulation consumes
                    the ticket counters of its binning"""
                    ts = []
                    for k in range(warm + iters):
                        bin_()
                        a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        a_.record(); fn(); b_.record()
                        torch.cuda.synchronize()
                        if k >= warm:
                            ts.append(a_.elapsed_time(b_) * 1e3)
                    return sorted(ts)[len(ts) // 2]

                bin_()
                r = {"bin_us": median_us(bin_, 5, 30), "accumulate_us": timed(acc), "accumulate_adam_us": timed(acc_adam)}
                if sname in ("default", "build_default", "dealt"):
                    r["accumulate_bf16_2groups_us"] = timed(acc_bf16([(11, 16), (0, 11)]))
                    r["accumulate_bf16_hi_only_us"] = timed(acc_bf16([(11, 16)]))
                bin_()
                acc()
                torch.cuda.synchronize()
                gd64 = g.double()
                r["grad_norm"] = float(gd64.norm())
                r["level_sums"] = [float(gd64[off[l]:off[l + 1]].sum()) for l in range(16)]
                r["level_abs_sums"] = [float(gd64[off[l]:off[l + 1]].abs().sum()) for l in range(16)]
                res[f"{cfg_name}:{label}"] = {k: (round(val, 2) if k.endswith("_us") else val) for k, val in r.items()}
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    if os.environ.get("NSR_VARIANT_WORKER"):
        worker()
    else:
        for so in sys.argv[1:]:
            env = dict(os.environ, NSR_HIP_LIB=os.path.abspath(so), NSR_VARIANT_WORKER="1")
            p = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
            print(p.stdout.strip() or json.dumps({"lib": so, "error": p.stderr[-600:]}), flush=True)
