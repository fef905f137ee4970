"""A/B of the hash-grid forward decompositions the north-star names (LDS-staged per-level tiles) against the shipped kernel:
E2 ray-coherent and E1 uniform inputs, C2 shapes, N = 2^18 and the bench's marched-sample size.  Prints one JSON; with
NSR_AB_VARIANT="lds,lpl" it only runs that variant a few hundred times (for `rocprofv3 --pmc TCC_REQ_sum TCP_TCC_READ_REQ_sum`).

    python tools/forward_ab.py > profiles/rNN_forward_ab.json
"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd"), os.path.join(ROOT, "tools")]
import torch
import nsr_hip
from nsr_hip import check, lib, ops
from kernel_microbench import coherent, median_us  # noqa: E402

gd = nsr_hip.make_grid_desc(16, 2, 19, 16, 1.447269237440378)
torch.manual_seed(1337)
table = (torch.randn(gd.n_entries * 2, device="cuda") * 0.1).half()
VARIANTS = [(0, 1), (1, 1), (2, 1), (0, 2), (2, 2)]
only = os.environ.get("NSR_AB_VARIANT")
res = {"shapes": "HashGrid L16 T2^19 F2 (levels 0/1: 4096 / 13824 entries = 16 / 54 KB, the only ones that fit 64 KB of LDS)",
       "variants": "(lds_levels, levels_per_lane); (0, 1) = round-1 kernel, (0, 2) = shipped default", "cases": []}
for n, dist in ((1 << 18, "E2_coherent"), (1 << 18, "E1_uniform"), (280000, "E2_coherent")):
    x = torch.rand(n, 3, device="cuda") if dist == "E1_uniform" else coherent(n - n % 64)
    n = x.shape[0]
    y = torch.empty(n, 32, dtype=torch.float16, device="cuda")
    check(lib.nsr_hashgrid_forward_variant(0, 1))
    want = ops.hashgrid_forward(x, table, gd).clone()
    row = {"n": n, "inputs": dist}
    for lds, lpl in VARIANTS:
        if only and only != f"{lds},{lpl}":
            continue
        check(lib.nsr_hashgrid_forward_variant(lds, lpl))
        got = ops.hashgrid_forward(x, table, gd, out=y)
        assert torch.equal(got, want), (lds, lpl)
        row[f"lds{lds}_lpl{lpl}_us"] = median_us(lambda: ops.hashgrid_forward(x, table, gd, out=y), 20, 100 if not only else 300)
    res["cases"].append(row)
check(lib.nsr_hashgrid_forward_variant(0, 2))  # the shipped default
print(json.dumps(res, indent=1))
