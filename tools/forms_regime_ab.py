"""A/B of the step's forms AT FIXED POINTS OF TRAINING: a fresh model + trainer per form setting (same seed: the same trajectory
up to the step's run-to-run noise), windows timed at the steps bench.py reports on -- the driver's window (steps 305-325 of the
model), its default window (320-520), steady state (600-800), 1,500-1,700 -- in the order A B B A.  One JSON line.
    python tools/forms_regime_ab.py [name=keys:defer_pack:defer_weights:rpw:cap ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd")]
import torch
import nsr
from nsr.scene import SyntheticBlender
from nsr.trainer import Trainer, ROUND4_FORMS, CURRENT_FORMS, set_step_forms

dev = torch.device("cuda", 0)
cfg = nsr.configs.get("nerf-blender")
data = SyntheticBlender(n_images=100, w=800, h=800, device=dev, seed=0)
FORMS = {"round4_forms": ROUND4_FORMS, "current_forms": CURRENT_FORMS}
for spec in sys.argv[1:]:
    name, rest = spec.split("=")
    keys, dp, dw, cap = rest.split(":")  # e.g. mine=101:1:1:128 -> keys 0, 2, 5
    FORMS[name] = dict(keys=dict(zip((0, 2, 5), (int(c) for c in keys))), defer_pack=bool(int(dp)),
                       defer_weights_wait=bool(int(dw)), wgrad_max_blocks=int(cap))
WINDOWS = [(305, 325), (325, 525), (600, 800), (1500, 1700)]
# box warm-up (a throwaway model), as bench.py does
torch.manual_seed(1)
tmp = Trainer(nsr.build(cfg).to(dev).train(), data, cfg, seed=1, async_mode=True)
for _ in range(1000):
    tmp.train_step()
torch.cuda.synchronize()
del tmp
names = [n for n in FORMS]
order = names + names[::-1]
res = {n: {f"{a}-{b}": [] for a, b in WINDOWS} for n in names}
for name in order:
    torch.manual_seed(42)
    tr = Trainer(nsr.build(cfg).to(dev).train(), data, cfg, seed=42, async_mode=True)
    set_step_forms(tr, FORMS[name])
    for a, b in WINDOWS:
        while tr.global_step < a:
            tr.train_step()
        torch.cuda.synchronize()
        c0, t0 = tr.counters(), time.perf_counter()
        for _ in range(b - a):
            tr.train_step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        c1 = tr.counters()
        res[name][f"{a}-{b}"].append((round(1e3 * dt / (b - a), 4), round((c1["samples"] - c0["samples"]) / (b - a))))
    del tr
    torch.cuda.empty_cache()
set_step_forms(type("T", (), {"settle": lambda s: None, "fused": type("F", (), {})(), "defer_weights_wait": True})(), CURRENT_FORMS)
print(json.dumps({"windows": res, "forms": {k: {kk: (list(vv) if isinstance(vv, tuple) else vv) for kk, vv in v.items()} for k, v in FORMS.items()}}))
