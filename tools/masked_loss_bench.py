"""GPU time of F.smooth_l1_loss(pred[valid], target[valid]) + backward on 8,192 x 3 values: the gathered form (torch), the
deferred selections with torch's elementwise kernels (NSR_MASKED_LOSS_TORCH), the deferred selections on nsr_masked_loss_*."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd")]
import torch
import torch.nn.functional as F
from nsr.models import _ValidMask

g = torch.Generator().manual_seed(0)
pred = torch.rand(8192, 3, generator=g).cuda().requires_grad_(True)
target = torch.rand(8192, 3, generator=g).cuda()
mask = (torch.rand(8192, generator=g) < 0.8).cuda()
valid = mask.clone().as_subclass(_ValidMask)


def run(form, reps=300):
    os.environ.pop("NSR_MASKED_LOSS_TORCH", None)
    if form == "deferred_torch":
        os.environ["NSR_MASKED_LOSS_TORCH"] = "1"
    v = mask if form == "gathered" else valid
    for _ in range(20):
        pred.grad = None
        F.smooth_l1_loss(pred[v], target[v]).backward()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        pred.grad = None
        F.smooth_l1_loss(pred[v], target[v]).backward()
    e1.record()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    return {"gpu_us": 1e3 * e0.elapsed_time(e1) / reps, "host_us": 1e6 * th / reps}


print(json.dumps({f: run(f) for f in ("gathered", "deferred_torch", "deferred_kernels", "deferred_torch", "deferred_kernels")}))
