"""Same-process A/B of the round-5 forms of the NeRF step (nsr_nerf_step_variant + FusedNeRFStep.defer_pack): ONE trainer is
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

# name -> (variant keys 0..8, defer_pack, defer_weights_wait, rays per wave of the flat compositing, wgrad block cap)
# keys: 0 pair dgrad, 1 dense levels through atomics, 2 flat compositing, 3 two wgrad streams, 4 wgrads behind the table backward,
#       5 fork events ride on kernels, 6 events with a device-scope release, 7 table backward issued first, 8 pipelined half encodes
SETTINGS = {
    "round4_forms": ((0, 0, 0, 0, 0, 0, 0, 0, 0), False, False, 4, 512),
    "pair_only": ((1, 0, 0, 0, 0, 0, 0, 0, 0), False, False, 4, 512),
    "flat_only": ((0, 0, 1, 0, 0, 0, 0, 0, 0), False, False, 4, 512),
    "defer_pack_only": ((0, 0, 0, 0, 0, 0, 0, 0, 0), True, False, 4, 512),
    "defer_weights_only": ((0, 0, 0, 0, 0, 0, 0, 0, 0), False, True, 4, 512),
    "round5_forms": ((1, 0, 1, 0, 0, 1, 0, 0, 0), True, True, 4, 128),
    "round6_sample_partitioned_compositing": ((1, 0, 2, 0, 0, 1, 0, 0, 0), True, True, 4, 128),
    "round5_without_pair": ((0, 0, 1, 0, 0, 1, 0, 0, 0), True, True, 4, 128),
    "round5_plus_dense_atomics": ((1, 1, 1, 0, 0, 1, 0, 0, 0), True, True, 4, 128),
    "round5_plus_pipelined_encode": ((1, 0, 1, 0, 0, 1, 0, 0, 1), True, True, 4, 128),
    "round5_cap256": ((1, 0, 1, 0, 0, 1, 0, 0, 0), True, True, 4, 256),
    "round5_cap512": ((1, 0, 1, 0, 0, 1, 0, 0, 0), True, True, 4, 512),
    "round5_without_pair_cap512": ((0, 0, 1, 0, 0, 1, 0, 0, 0), True, True, 4, 512),
    "round5_two_wgrad_streams_cap512": ((1, 0, 1, 1, 0, 1, 0, 0, 0), True, True, 4, 512),
    # (sixth field: nsr_nerf_sigma_mode -- 1 = the ray-ordered sigma pass that stops at each ray's transmittance cut)
    "round5_sigma_rays": ((1, 0, 1, 0, 0, 1, 0, 0, 0), True, True, 4, 128, 1),
    # (key 10: the table backward on the helper stream behind its binning, weight gradients + MLP optimizer on the step's stream)
    "round5_table_on_helper": ((1, 0, 1, 0, 0, 1, 0, 0, 0, 0, 1), True, True, 4, 128),
    "round5_table_on_helper_cap512": ((1, 0, 1, 0, 0, 1, 0, 0, 0, 0, 1), True, True, 4, 512),
}
HOST_DELAY = float(os.environ.get("NSR_HOST_DELAY_US", "0")) * 1e-6
only = os.environ.get("NSR_VARIANTS")
if only:
    SETTINGS = {k: v for k, v in SETTINGS.items() if k in only.split(",")}


def apply(keys, defer, defer_w, rpw, cap=512, sigma_mode=0):
    tr.settle()
    torch.cuda.synchronize()
    lib.nsr_nerf_sigma_mode(sigma_mode)
    lib.nsr_nerf_step_variant(10, 0)
    for k, v in enumerate(keys):
        if k != 9:
            lib.nsr_nerf_step_variant(k, v)
    lib.nsr_nerf_step_variant(9, 0 if cap >= 512 else cap)
    lib.nsr_composite_flat_rays_per_wave(rpw)
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
apply((1, 0, 1, 0, 0, 1, 0, 0, 0), True, True, 4, 128)
out = {"train_steps": n_train, "timed_steps": n_timed, "rounds": rounds, "global_step": tr.global_step,
       "settings": {k: {"ms_per_step": [round(x["ms_per_step"], 4) for x in v],
                        "host_ms_per_step": [round(x["host_ms_per_step"], 4) for x in v],
                        "kept_per_step": [round(x["kept_per_step"]) for x in v],
                        "marched_per_step": [round(x["marched_per_step"]) for x in v],
                        "loss": [x["loss"] for x in v],
                        "mean_ms": round(sum(x["ms_per_step"] for x in v) / len(v), 4)} for k, v in res.items()}}
print(json.dumps(out))
