"""Where a step through nsr.models.FusedNeRFModel (bench.py boundary_path) spends its time: every phase bracketed by a
synchronize (so each figure = host + GPU time of that phase, no overlap), mean over 100 steps after 300 warm-up steps."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd")]
import torch
import nsr, nsr.models
from nsr.scene import SyntheticBlender

dev = torch.device("cuda", 0)
torch.manual_seed(42)
cfg = nsr.configs.get("nerf-blender")
model = nsr.models.FusedNeRFModel(cfg).to(dev).train()
data = SyntheticBlender(n_images=100, w=800, h=800, device=dev, seed=0)
gen = torch.Generator(device=dev); gen.manual_seed(42)
opt = torch.optim.AdamW(model.parameters(), lr=0.01, betas=(0.9, 0.99), eps=1e-15, fused=True)
n_rays, target = cfg["train_num_rays"], cfg["train_num_rays"] * cfg["num_samples_per_ray"]
acc = {}
NOSYNC = bool(os.environ.get("NSR_BP_NOSYNC"))  # plain loop (for a rocprofv3 kernel trace of the un-instrumented step)
def lap(name, t0):
    if NOSYNC:
        return t0
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t1 - t0)
    return t1
for step in range(400):
    prof = step >= 300
    if not NOSYNC:
        torch.cuda.synchronize()
    t = time.perf_counter()
    rays, rgb, fg, bg = data.sample_rays(n_rays, gen, cfg["background_color"])
    if prof: t = lap("sample_rays", t)
    model.background_color = bg
    model.update_step(0, step)
    if prof: t = lap("update_step", t)
    out = model(rays)
    if prof: t = lap("forward", t)
    n = int(out["num_samples"].sum().item())
    tt = int(n_rays * (target / max(n, 1)))
    n_rays = min(int(n_rays * 0.9 + tt * 0.1), cfg["max_train_num_rays"])
    valid = out["rays_valid"][..., 0]
    loss = torch.nn.functional.smooth_l1_loss(out["comp_rgb"][valid], rgb[valid])
    if prof: t = lap("loss", t)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    if prof: t = lap("backward", t)
    opt.step()
    if prof: t = lap("optimizer", t)
print(json.dumps({k: round(1e3 * v / 100, 4) for k, v in acc.items()} | {"kept": n, "rays": n_rays}))
