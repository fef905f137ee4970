"""A/B of the one-kernel encode -> MLP forward (csrc/gridmlp.hip) against the XCD-placed encode + MLP pair, nerf-blender
density network (L16 T2^19 F2 -> 64 -> 16), ray-coherent and uniform positions, inference (only the outputs leave) and
training (activations + encoded features saved).  One JSON object on stdout.

    python tools/grid_mlp_ab.py > gpurun_out/grid_mlp_ab.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd"), os.path.join(ROOT, "tools")]
import torch  # noqa: E402
import nsr_hip  # noqa: E402
from nsr_hip import lib, ops  # noqa: E402
from kernel_microbench import coherent, median_us  # noqa: E402

gd = nsr_hip.make_grid_desc(16, 2, 19, 16, 1.447269237440378)
res = {"grid": "L16 T2^19 F2", "cases": []}
for n_hidden in (1, 2):
    md = nsr_hip.NsrMlpDesc(32, 32, 16, 16, n_hidden, 0)
    g = torch.Generator().manual_seed(0)
    table = ((torch.rand(gd.n_entries * 2, generator=g) * 2 - 1) * 0.1).half().cuda()
    w = (torch.randn(64 * 32 + (n_hidden - 1) * 4096 + 1024, generator=g) * 0.1).half().cuda()
    for kind in ("coherent", "uniform"):
        for n in (2048, 8192, 32768, 65536, 131072, 262144, 1048576):
            x = coherent((n + 63) // 64 * 64, per_ray=64)[:n].contiguous() if kind == "coherent" else torch.rand(n, 3, device="cuda")
            row = {"n_hidden": n_hidden, "inputs": kind, "n": n}
            for train in (False, True):
                tag = "train" if train else "infer"

                def pair():
                    enc = ops.hashgrid_forward(x, table, gd)
                    ops.mlp_forward(enc, w, md, save_acts=train)

                row[f"pair_{tag}_us"] = round(median_us(pair), 2)
                # (the grid-size cap is fixed at 512 since round 6: profiles/r03_grid_mlp_ab.json has the 512 / 2048 / 8192 sweep)
                row[f"fused_{tag}_b512_us"] = round(median_us(
                    lambda: ops.grid_mlp_forward(x, table, w, gd, md, save_acts=train, want_enc=train)), 2)
            res["cases"].append(row)
            print(json.dumps(row), file=sys.stderr)
print(json.dumps(res))
