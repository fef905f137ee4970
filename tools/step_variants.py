"""Same-process A/B of the forms of the NeRF step (nsr_nerf_step_variant + FusedNeRFStep.defer_pack): ONE trainer is
brought to a regime (steady ~step 700, late ~step 10,000), then windows of `timed` steps are run with each setting in turn,
interleaved over `rounds` rounds so that the slow drift of the sample counts hits every setting alike.
    python tools/step_variants.py [train_steps] [timed_steps] [rounds]  -> one JSON line"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd")]
import torch
import nsr
from nsr.scene import SyntheticBlender
from nsr.trainer import Trainer
from nsr_hip import lib

n_train = int(sys.argv[1]) if len(sys.argv) > 1 else 700
n_timed = int(sys.argv[2]) if len(sys.argv) > 2 else 160  # multiple of 16: every window holds the same number of grid refreshes
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda", 0)
torch.manual_seed(42)
cfg = nsr.configs.get("nerf-blender")
model = nsr.build(cfg).to(dev).train()
data = SyntheticBlender(n_images=int(os.environ.get("NSR_LATE_IMAGES", "100")), w=400, h=400, device=dev, seed=0)
tr = Trainer(model, data, cfg, seed=42, async_mode=True)

# name -> (nsr_nerf_step_variant keys {0: pair dgrad, 2: sample-partitioned compositing, 5: fork events ride on kernels},
#          defer_pack, defer_weights_wait, wgrad block cap)
SETTINGS = {
    "round4_forms": ({0: 0, 2: 0, 5: 0}, False, False, 512),
    "current_forms": ({0: 1, 2: 1, 5: 1}, True, True, 128),
    "current_wave_per_ray_compositing": ({0: 1, 2: 0, 5: 1}, True, True, 128),
    "current_without_pair": ({0: 0, 2: 1, 5: 1}, True, True, 128),
    "current_cap512": ({0: 1, 2: 1, 5: 1}, True, True, 512),
}
HOST_DELAY = float(os.environ.get("NSR_HOST_DELAY_US", "0")) * 1e-6
only = os.environ.get("NSR_VARIANTS")
if only:
    SETTINGS = {k: v for k, v in SETTINGS.items() if k in only.split(",")}


def apply(keys, defer, defer_w, cap=512):
    tr.settle()
    torch.cuda.synchronize()
    for k, v in keys.items():
        lib.nsr_nerf_step_variant(k, v)
    lib.nsr_nerf_step_variant(9, 0 if cap >= 512 else cap)
    tr.fused.defer_pack = defer
    tr.defer_weights_wait = defer_w


for _ in range(n_train):
    tr.train_step()
torch.cuda.synchronize()
res = {k: [] for k in SETTINGS}
names = list(SETTINGS)
for r in range(rounds):
    order = names[r % len(names):] + names[:r % len(names)]  # rotated: the drift of the sample counts hits every setting alike
    if r % 2:
        order = order[::-1]
    for name in order:
        apply(*SETTINGS[name])
        for _ in range(16):
            tr.train_step()
        torch.cuda.synchronize()
        c0 = tr.counters()
        t0 = time.perf_counter()
        for _ in range(n_timed):
            tr.train_step()
            if HOST_DELAY > 0:  # (experiment: is the step's critical path coupled to the host's enqueue time?)
                td = time.perf_counter() + HOST_DELAY
                while time.perf_counter() < td:
                    pass
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        c1 = tr.counters()
        res[name].append({"ms_per_step": 1e3 * (t2 - t0) / n_timed, "host_ms_per_step": 1e3 * (t1 - t0) / n_timed,
                          "kept_per_step": (c1["samples"] - c0["samples"]) / n_timed,
                          "marched_per_step": (c1["marched"] - c0["marched"]) / n_timed,
                          "loss": float(tr.last["loss"])})
apply(*SETTINGS["current_forms"])
out = {"train_steps": n_train, "timed_steps": n_timed, "rounds": rounds, "global_step": tr.global_step,
       "settings": {k: {"ms_per_step": [round(x["ms_per_step"], 4) for x in v],
                        "host_ms_per_step": [round(x["host_ms_per_step"], 4) for x in v],
                        "kept_per_step": [round(x["kept_per_step"]) for x in v],
                        "marched_per_step": [round(x["marched_per_step"]) for x in v],
                        "loss": [x["loss"] for x in v],
                        "mean_ms": round(sum(x["ms_per_step"] for x in v) / len(v), 4)} for k, v in res.items()}}
print(json.dumps(out))
