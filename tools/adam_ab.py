import sys, os, torch
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "instant-nsr-pl_amd")]
from nsr_hip import ops
n = 12602992
def mk(n):
    p = torch.randn(n, device="cuda") * 0.1
    return [p, torch.randn_like(p), torch.zeros_like(p), torch.zeros_like(p), torch.empty(n, dtype=torch.float16, device="cuda")]
a, b = mk(n), mk(7168)
step = torch.zeros(1, dtype=torch.int32, device="cuda"); hyper = torch.zeros(12, device="cuda")
def sep():
    ops.adam_tick(step, hyper, 0.01, 0.9, 0.99, 0.33, (10000, 15000, 18000))
    ops.adamw_step(*a, 0.01, 0.9, 0.99, 1e-15, 0.01, 1, zero_grad=True, hyper=hyper, zero_first_n=3072)
    ops.adamw_step(*b, 0.01, 0.9, 0.99, 1e-15, 0.01, 1, zero_grad=True, hyper=hyper)
def fused():
    ops.adamw_step_scheduled([tuple(a) + (3072,), tuple(b) + (0,)], step, hyper, 0.01, 0.9, 0.99, 0.33, (10000, 15000, 18000), 1e-15, 0.01)
for name, fn in (("separate", sep), ("scheduled", fused), ("separate", sep), ("scheduled", fused)):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): fn()
    e1.record(); torch.cuda.synchronize()
    print(name, e0.elapsed_time(e1) / 200 * 1e3, "us")
