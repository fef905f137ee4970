"""Stand-alone k_adamw sweep over a 12.6 M-entry fp32 vector (the hash table of nerf-blender): KNOWN traffic -- p, g, m, v read
(16 B / parameter), p, m, v + the fp16 image written and g zeroed (18 B / parameter).  Run under
    rocprofv3 --kernel-trace --pmc FETCH_SIZE   (and a second pass with WRITE_SIZE)
the counter averages of its dispatches calibrate the read- / write-side corrections that tools/pmc_traffic.py applies to the
table backward's counters (the x2 read correction was applied uncalibrated).  tools/fetch_calibration.sh drives it."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd")]
import torch
from nsr_hip import ops

n = 12599920
p = torch.randn(n, device="cuda") * 0.1
g = torch.randn(n, device="cuda") * 1e-3
m, v = torch.zeros_like(p), torch.zeros_like(p)
h = torch.empty(n, dtype=torch.float16, device="cuda")
for it in range(40):
    ops.adamw_step(p, g, m, v, h, 0.01, 0.9, 0.99, 1e-15, 0.01, it + 1, zero_grad=True)
    g.normal_(0, 1e-3)  # (a second, smaller kernel: the summary picks k_adamw by name)
torch.cuda.synchronize()
print("ok", n)
