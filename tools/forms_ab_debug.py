"""why does bench.py's step_forms_ab see no gain where tools/step_variants.py sees 10 %?  Same A/B under bench's conditions, one
factor at a time: image size (800 vs 400), a throwaway trainer before (burn-in), HIP-event profiling legs before."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd")]
import torch
import nsr
from nsr.scene import SyntheticBlender
from nsr.trainer import Trainer, ROUND4_FORMS, ROUND5_FORMS, set_step_forms
from nsr_hip import ops

res_px, burn, prof, n_train = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dev = torch.device("cuda", 0)
cfg = nsr.configs.get("nerf-blender")
data = SyntheticBlender(n_images=100, w=res_px, h=res_px, device=dev, seed=0)
if burn:
    torch.manual_seed(1)
    tmp = Trainer(nsr.build(cfg).to(dev).train(), data, cfg, seed=1, async_mode=True)
    for _ in range(1500):
        tmp.train_step()
    torch.cuda.synchronize()
    del tmp
    torch.cuda.empty_cache()
torch.manual_seed(42)
tr = Trainer(nsr.build(cfg).to(dev).train(), data, cfg, seed=42, async_mode=True)
for _ in range(n_train):
    tr.train_step()
if prof:
    ops.profile_begin(native_only=True)
    for _ in range(64):
        tr.train_step()
    ops.profile_end()
    ops.profile_begin()
    for _ in range(32):
        tr.train_step()
    ops.profile_end()
    tr.fuse_table_update = False
    for _ in range(32):
        tr.train_step()
    tr.fuse_table_update = True
torch.cuda.synchronize()
acc = {"round4_forms": [], "round5_forms": []}
for name in ("round4_forms", "round5_forms", "round5_forms", "round4_forms", "round4_forms", "round5_forms"):
    set_step_forms(tr, ROUND4_FORMS if name == "round4_forms" else ROUND5_FORMS)
    for _ in range(16):
        tr.train_step()
    torch.cuda.synchronize()
    c0, t0 = tr.counters(), time.perf_counter()
    for _ in range(160):
        tr.train_step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c1 = tr.counters()
    acc[name].append((round(1e3 * dt / 160, 4), round((c1["samples"] - c0["samples"]) / 160)))
print(json.dumps({"res": res_px, "burn_in": burn, "profile_legs": prof, "n_train": n_train, **acc}))
