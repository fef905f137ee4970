"""End-to-end soak of the fused NeuS / neuralangelo step on the procedural scene: the reference's schedule (losses and LR
schedule of the YAML, dynamic ray count, occupancy refresh every 16 steps, cos-anneal, progressive levels) for --steps
steps, then PSNR of unseen views rendered in eval mode through nsr.export.render_rays.  One JSON line.

    python tools/train_neus.py --config neus-blender|neus-dtu|neuralangelo --steps 4000
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd")]
import torch
import nsr
from nsr.export import render_rays
from nsr.fused_neus import NeuSTrainer
from nsr.scene import SyntheticBlender, get_rays

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="neus-blender")
ap.add_argument("--steps", type=int, default=4000)
ap.add_argument("--test-views", type=int, default=3)
ap.add_argument("--res", type=int, default=200)
args = ap.parse_args()
torch.manual_seed(42)
dev = "cuda"
cfg = nsr.configs.get(args.config)
LAM = {"neus-blender": {"lambda_rgb_mse": 10.0, "lambda_rgb_l1": 0.0, "lambda_mask": 0.1, "lambda_eikonal": 0.1},
       "neus-dtu": {"lambda_rgb_mse": 0.0, "lambda_rgb_l1": 1.0, "lambda_mask": 0.0, "lambda_eikonal": 0.1},
       "neuralangelo": {"lambda_rgb_mse": 0.0, "lambda_rgb_l1": 1.0, "lambda_mask": 0.1, "lambda_eikonal": 0.1}}[args.config]
scale = float(cfg["radius"]) / 1.5
env = bool(cfg["learned_background"])  # unmasked captures with a backdrop: the target of the NeRF++ background branch
train = SyntheticBlender(n_images=100, w=args.res, h=args.res, device=dev, seed=0, environment=env)
test = SyntheticBlender(n_images=args.test_views, w=args.res, h=args.res, device=dev, seed=12345, environment=env)
for d in (train, test):
    d.all_c2w[:, :, 3] *= scale  # the procedural object lives in radius 1.5: shrink the camera orbit with the box ...
model = nsr.build(cfg).to(dev).train()
tr = NeuSTrainer(model, train, cfg, LAM, config_name=args.config, max_steps=max(args.steps, 1000))
if args.config == "neuralangelo":
    cfg["geometry"]["xyz_encoding_config"]["update_steps"] = max(args.steps // 16, 1)  # reach all 16 levels within the soak
    model.progressive["update_steps"] = cfg["geometry"]["xyz_encoding_config"]["update_steps"]
torch.cuda.synchronize(); t0 = time.perf_counter()
n_samples, hist = 0, []
for i in range(args.steps):
    st = tr.train_step()
    n_samples += st["n_samples"] + st["n_samples_bg"]
    if (i + 1) % max(args.steps // 8, 1) == 0:
        terms = tr.fused.loss_terms(st["loss_acc"])
        hist.append({"step": i + 1, "rays": st["n_rays"], "samples": st["n_samples"],
                     **{k: round(float(v), 5) for k, v in terms.items() if k in ("rgb_l1", "rgb_mse", "eikonal", "mask")}})
torch.cuda.synchronize(); dt = time.perf_counter() - t0
model.eval()
psnrs = []
for i in range(args.test_views):
    o, d = get_rays(test.directions.view(-1, 3), test.all_c2w[i:i + 1].expand(args.res * args.res, -1, -1))
    rays = torch.cat([o, torch.nn.functional.normalize(d, p=2, dim=-1)], -1)
    out = render_rays(tr.fused, rays, chunk=16384)
    fg = test.all_fg_masks[i].view(-1, 1).cpu()
    gt = test.all_images[i].view(-1, 3).cpu() if env else test.all_images[i].view(-1, 3).cpu() * fg + (1 - fg)
    mse = torch.mean((out["comp_rgb_full"].clamp(0, 1) - gt) ** 2)
    psnrs.append(float(-10.0 * torch.log10(mse)))
print(json.dumps({"config": args.config, "steps": args.steps, "train_seconds": dt, "ms_per_step": 1e3 * dt / args.steps,
                  "samples_per_sec": n_samples / dt, "test_psnr": sum(psnrs) / len(psnrs), "inv_s": float(model.variance.variance.exp().pow(10)),
                  "history": hist}))
