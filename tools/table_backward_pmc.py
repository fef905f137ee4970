"""Workload for an isolated PMC pass over the table backward (run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and
again with WRITE_SIZE): for each placement of the work units (listed / dealt) and each operating point of bench.py
(9.6e4 / 2.16e5 ray-coherent samples): one binning launch, then 6 accumulate + AdamW launches.  The dispatch ORDER is what
tools/table_backward_pmc_summary.py keys on.

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pt -o t -- python tools/table_backward_pmc.py
"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd"), os.path.join(ROOT, "tools")]
import torch
import nsr_hip
from nsr_hip import check, lib, ptr, stream_ptr
from kernel_microbench import coherent

CASES = [(placement, n) for placement in (0, 2, 1) for n in (96000, 216000)]
REPS = 6

if __name__ == "__main__":
    gd = nsr_hip.make_grid_desc(16, 2, 19, 16, 1.447269237440378)
    P = gd.n_entries * 2
    D = ctypes.byref(gd)
    lib.nsr_hashgrid_owner_large_from(0xffffffff)
    p, m, v = torch.randn(P, device="cuda") * 1e-2, torch.zeros(P, device="cuda"), torch.zeros(P, device="cuda")
    sh = torch.empty(P, dtype=torch.float16, device="cuda")
    step, hyper = torch.zeros(1, dtype=torch.int32, device="cuda"), torch.zeros(12, device="cuda")
    ad = nsr_hip.NsrTableAdam()
    ad.params, ad.exp_avg, ad.exp_avg_sq, ad.shadow = p.data_ptr(), m.data_ptr(), v.data_ptr(), sh.data_ptr()
    ad.step, ad.hyper = step.data_ptr(), hyper.data_ptr()
    ad.base_lr, ad.beta1, ad.beta2, ad.gamma = 0.01, 0.9, 0.99, 0.33
    ad.milestone0, ad.milestone1, ad.milestone2 = 10000, 15000, 18000
    ad.eps, ad.weight_decay = 1e-15, 0.01
    for placement, n in CASES:
        lib.nsr_hashgrid_owner_tune(0, float(placement))
        x = coherent((n + 63) // 64 * 64, per_ray=16)[:n].contiguous()
        dy = torch.randn(16, n, 2, device="cuda") * 1e-3
        ws = torch.empty(int(lib.nsr_hashgrid_backward_params_workspace_floats(D, n)), device="cuda")
        big = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
        for _ in range(REPS):
            big.zero_()  # push x / dy / the table out of the L2s and most of the Infinity Cache between launches
            check(lib.nsr_hashgrid_backward_params_owner_bin(ptr(x), ptr(ws), n, 16, D, None, stream_ptr()), "bin")
            check(lib.nsr_hashgrid_backward_params_owner_accumulate_adam(ptr(x), ptr(dy), 2, 0, ptr(ws), n, 16, 1.0, D, None,
                                                                         ctypes.byref(ad), stream_ptr()), "acc_adam")
        torch.cuda.synchronize()
