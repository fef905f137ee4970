"""Where the HOST time of an asynchronous NeRF step goes: every entry point of libnsr_hip the step calls is wrapped with a
timer (time inside the C call = HIP launches + event calls), torch's event / stream methods likewise; the rest is Python.
    python tools/host_breakdown.py [train_steps] [timed_steps]"""
import collections, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd")]
import torch
import nsr, nsr_hip
from nsr.scene import SyntheticBlender
from nsr.trainer import Trainer

n_train = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
n_timed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dev = torch.device("cuda", 0)
torch.manual_seed(42)
cfg = nsr.configs.get("nerf-blender")
model = nsr.build(cfg).to(dev).train()
data = SyntheticBlender(n_images=24, w=400, h=400, device=dev, seed=0)
tr = Trainer(model, data, cfg, seed=42, async_mode=True)
for _ in range(n_train):
    tr.train_step()
torch.cuda.synchronize()
acc, calls = collections.defaultdict(float), collections.defaultdict(int)
pc = time.perf_counter


def wrap(name, fn):
    def w(*a, **k):
        t = pc()
        r = fn(*a, **k)
        acc[name] += pc() - t
        calls[name] += 1
        return r
    return w


lib = nsr_hip.lib
for name in list(nsr_hip.SIGNATURES):
    setattr(lib, name, wrap("C:" + name, getattr(lib, name)))
for cls, meths in ((torch.cuda.Event, ("record", "wait")), (torch.cuda.Stream, ("wait_event", "record_event"))):
    for m in meths:
        setattr(cls, m, wrap(f"torch:{cls.__name__}.{m}", getattr(cls, m)))
torch.cuda.current_stream = wrap("torch:current_stream", torch.cuda.current_stream)
t0 = pc()
for _ in range(n_timed):
    tr.train_step()
t1 = pc()
torch.cuda.synchronize()
total = 1e6 * (t1 - t0) / n_timed
rows = sorted(((1e6 * v / n_timed, calls[k] / n_timed, k) for k, v in acc.items()), reverse=True)
inside = sum(r[0] for r in rows)
print(json.dumps({"host_us_per_step": round(total, 1), "inside_wrapped_calls_us": round(inside, 1),
                  "python_remainder_us": round(total - inside, 1),
                  "calls": [{"us": round(u, 2), "per_step": round(n, 2), "what": k} for u, n, k in rows if u > 0.3]}))
