"""Timing of the NeuS hot path (C3 shapes, configs/neus-blender.yaml) through the DROP-IN packages -- the way the
reference's own models/neus.py would drive them: NeuSModel.forward_ (march, hash encode + VanillaMLP SDF with analytic
normals via double backward, colour MLP, get_alpha, alpha compositing) + rgb / eikonal loss + backward + AdamW.
Not the headline benchmark (bench.py); one JSON line for DESIGN.md.

    python tools/neus_step_bench.py [--rays 4096] [--steps 30]
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "instant-nsr-pl_amd"))
import torch
import nsr
from nsr.scene import SyntheticBlender

ap = argparse.ArgumentParser()
ap.add_argument("--rays", type=int, default=4096)
ap.add_argument("--steps", type=int, default=30)
args = ap.parse_args()
torch.manual_seed(0)
dev = "cuda"
cfg = nsr.configs.get("neus-blender")
model = nsr.NeuSModel(cfg).to(dev).train()
data = SyntheticBlender(n_images=20, w=400, h=400, device=dev, seed=0)
opt = torch.optim.AdamW(model.parameters(), lr=0.01, betas=(0.9, 0.99), eps=1e-15)
gen = torch.Generator(device=dev).manual_seed(1)
n_samples = 0


def step(i):
    global n_samples
    model.update_step(0, i)
    rays, rgb, fg, bg = data.sample_rays(args.rays, gen, cfg["background_color"])
    model.background_color = bg
    out = model(rays)
    valid = out["rays_valid_full"][..., 0] if "rays_valid_full" in out else out["rays_valid"][..., 0]
    loss = torch.nn.functional.mse_loss(out["comp_rgb"][valid], rgb[valid]) * 10.0
    eik = ((torch.linalg.norm(out["sdf_grad_samples"], ord=2, dim=-1) - 1.0) ** 2).mean()   # systems/neus.py:106
    loss = loss + 0.1 * eik
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    n_samples += int(out["num_samples"].sum())
    return float(loss)


for i in range(16):
    step(i)
torch.cuda.synchronize(); n_samples = 0; t0 = time.perf_counter()
for i in range(16, 16 + args.steps):
    last = step(i)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(json.dumps({"config": "neus-blender (C3 shapes), modular drop-in path, analytic normals (double backward)",
                  "rays_per_step": args.rays, "steps": args.steps, "ms_per_step": 1e3 * dt / args.steps,
                  "samples_per_step": n_samples / args.steps, "samples_per_sec": n_samples / dt, "last_loss": last}))
