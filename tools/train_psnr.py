"""End-to-end quality check of the measured path: train the nerf-blender config for N steps with the asynchronous fused
step on the procedural scene, then render held-out views with the eval path (chunk_batch, models/nerf.py:111-127 +
systems/nerf.py:118-160 semantics: white background, PSNR on the masked-composited image) and print one JSON line.

    python tools/train_psnr.py [--steps 20000] [--test-views 4] [--res 400]
"""
import argparse, json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "instant-nsr-pl_amd"))
import torch
import nsr
from nsr.scene import SyntheticBlender, get_rays
from nsr.trainer import Trainer

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20000)
ap.add_argument("--test-views", type=int, default=4)
ap.add_argument("--res", type=int, default=400)
args = ap.parse_args()
torch.manual_seed(42)
dev = "cuda"
cfg = nsr.configs.get("nerf-blender")
model = nsr.build(cfg).to(dev).train()
train = SyntheticBlender(n_images=100, w=args.res, h=args.res, device=dev, seed=0)
test = SyntheticBlender(n_images=args.test_views, w=args.res, h=args.res, device=dev, seed=12345)  # unseen cameras
tr = Trainer(model, train, cfg, seed=42, async_mode=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(args.steps):
    tr.train_step()
c = tr.counters(); dt = time.perf_counter() - t0
final_loss = float(tr.last["loss"])
params_finite = all(bool(torch.isfinite(p).all()) for p in model.parameters())
model.eval()
from nsr.export import render_rays
psnrs = []
with torch.no_grad():
    for i in range(args.test_views):
        o, d = get_rays(test.directions.view(-1, 3), test.all_c2w[i:i + 1].expand(args.res * args.res, -1, -1))
        rays = torch.cat([o, torch.nn.functional.normalize(d, p=2, dim=-1)], -1)
        out = render_rays(tr.fused, rays)  # eval-mode chunked render (ray_chunk pieces, results on the CPU)
        fg = test.all_fg_masks[i].view(-1, 1)
        gt = test.all_images[i].view(-1, 3) * fg + (1 - fg)
        mse = torch.mean((out["comp_rgb"].to(dev).clamp(0, 1) - gt) ** 2)  # chunk_batch offloads to the CPU like the reference
        psnrs.append(float(-10.0 * torch.log10(mse)))
print(json.dumps({"steps": args.steps, "train_seconds": dt, "ms_per_step": 1e3 * dt / args.steps,
                  "samples_per_sec": c["samples"] / dt, "rays_per_sec": c["rays"] / dt, "truncated_launches": c["truncated"],
                  "final_train_loss": final_loss, "parameters_finite": params_finite,
                  "test_psnr": sum(psnrs) / len(psnrs), "test_psnr_per_view": psnrs, "test_res": args.res}))
