"""workload for the PMC passes of the encode -> MLP A/B: 30 launches of the pair and 30 of the fused kernel on one input
kind (argv[1]: coherent | uniform) at argv[2] samples (default 262144), inference.  Run under
rocprofv3 --kernel-trace --pmc FETCH_SIZE (tools/grid_mlp_pmc.sh)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd"), os.path.join(ROOT, "tools")]
import torch  # noqa: E402
import nsr_hip  # noqa: E402
from nsr_hip import ops  # noqa: E402
from kernel_microbench import coherent  # noqa: E402

kind, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 262144
gd = nsr_hip.make_grid_desc(16, 2, 19, 16, 1.447269237440378)
md = nsr_hip.NsrMlpDesc(32, 32, 16, 16, 1, 0)
g = torch.Generator().manual_seed(0)
table = ((torch.rand(gd.n_entries * 2, generator=g) * 2 - 1) * 0.1).half().cuda()
w = (torch.randn(64 * 32 + 1024, generator=g) * 0.1).half().cuda()
x = coherent(n, per_ray=64) if kind == "coherent" else torch.rand(n, 3, device="cuda")
for _ in range(30):
    ops.mlp_forward(ops.hashgrid_forward(x, table, gd), w, md, save_acts=False)
for _ in range(30):
    ops.grid_mlp_forward(x, table, w, gd, md)
torch.cuda.synchronize()
