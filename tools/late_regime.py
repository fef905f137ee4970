"""The asynchronous NeRF step LATE in training (sparse grid, ~4e4 kept samples per step): ms per step, the host's time to queue
a step, kept / marched samples per step -- the regime that sets the wall time of a 20,000-step run.
    python tools/late_regime.py [train_steps] [timed_steps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd")]
import torch
import nsr
from nsr.scene import SyntheticBlender
from nsr.trainer import Trainer

n_train = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
n_timed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dev = torch.device("cuda", 0)
torch.manual_seed(42)
cfg = nsr.configs.get("nerf-blender")
model = nsr.build(cfg).to(dev).train()
data = SyntheticBlender(n_images=int(os.environ.get("NSR_LATE_IMAGES", "100")), w=400, h=400, device=dev, seed=0)
tr = Trainer(model, data, cfg, seed=42, async_mode=True)
out = {}
done = 0
for target in (1000, n_train):
    while done < target:
        tr.train_step(); done += 1
    torch.cuda.synchronize()
    c0 = tr.counters()
    t0 = time.perf_counter()
    for _ in range(n_timed):
        tr.train_step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    done += n_timed
    c1 = tr.counters()
    out[str(target)] = {"ms_per_step": 1e3 * (t2 - t0) / n_timed, "host_enqueue_ms_per_step": 1e3 * (t1 - t0) / n_timed,
                        "kept_per_step": (c1["samples"] - c0["samples"]) / n_timed,
                        "marched_per_step": (c1["marched"] - c0["marched"]) / n_timed,
                        "rays_per_step": (c1["rays"] - c0["rays"]) / n_timed}
print(json.dumps(out))
