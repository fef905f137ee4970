"""How much of the sigma pass's work lies behind the visibility cut of its ray?  Trains the nerf-blender config to the
operating point, then reads one step's per-ray marched and kept counts: kept samples are a PREFIX of a ray's marched samples
(transmittance is monotone), so a ray-ordered evaluation that stops at the cut would touch ceil((kept + 1) / g) * g samples
per ray at granularity g.   python tools/early_exit_fraction.py [steps]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd")]
import torch
import nsr
from nsr.scene import SyntheticBlender
from nsr.trainer import Trainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 700
dev = torch.device("cuda", 0)
torch.manual_seed(42)
cfg = nsr.configs.get("nerf-blender")
model = nsr.build(cfg).to(dev).train()
data = SyntheticBlender(n_images=24, w=400, h=400, device=dev, seed=0)
tr = Trainer(model, data, cfg, seed=42, async_mode=True)
out = {}
for s in range(steps):
    tr.train_step()
    if s + 1 in (320, steps):
        torch.cuda.synchronize()
        a = tr._async_state()
        rs = a["sets3"][(tr.global_step - 1) % a["window"]]
        ab = tr.fused._ab
        slots = rs["slots"]
        marched = rs["packed"][:, 1].long()
        kept = ab["meta"][:slots].long()
        res = {"rays": int((marched > 0).sum()), "marched": int(marched.sum()), "kept": int(kept.sum()),
               "rays_cut_before_end": int((kept < marched).sum()),
               "marched_per_ray_hist": torch.bincount(torch.clamp((marched + 15) // 16, max=12), minlength=13).tolist()}
        for g in (8, 16, 32, 64):
            ev = torch.minimum(marched, (kept + 1 + g - 1) // g * g)
            res[f"evaluated_g{g}"] = int(ev.sum())
            res[f"fraction_g{g}"] = float(ev.sum()) / max(float(marched.sum()), 1.0)
            res[f"tiles_g{g}"] = int(((ev + g - 1) // g).sum())
        out[str(s + 1)] = res
print(json.dumps(out))
