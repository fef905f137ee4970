"""Timing of one NeuS / neuralangelo training step (march, hash encode, fp32 SDF network with analytic normals or the
7-point finite-difference stencil, colour network, SDF->alpha compositing, system losses, backward, AdamW):
the FUSED runner (nsr/fused_neus.py) and, for comparison, the modular path through the drop-in packages (autograd over
~250 launches, the way the reference's models/neus.py drives them).  One JSON line.

    python tools/neus_step_bench.py [--config neus-blender|neus-dtu|neuralangelo] [--rays 4096] [--steps 30] [--modular]
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd"), os.path.join(ROOT, "tests")]
import torch
import nsr
from nsr.scene import SyntheticBlender

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="neus-blender")
ap.add_argument("--rays", type=int, default=4096)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--warmup", type=int, default=16)
ap.add_argument("--level-step", type=int, default=12005, help="neuralangelo: global step that sets the progressive level")
ap.add_argument("--modular", action="store_true")
ap.add_argument("--dynamic", action="store_true", help="fused path: the reference's dynamic ray count (targets 2^18 samples/step)")
args = ap.parse_args()
torch.manual_seed(0)
dev = "cuda"
cfg = nsr.configs.get(args.config)
LAM = {"lambda_rgb_l1": 1.0, "lambda_mask": 0.1, "lambda_eikonal": 0.1}
data = SyntheticBlender(n_images=20, w=400, h=400, device=dev, seed=0, environment=bool(cfg["learned_background"]))
scale = float(cfg["radius"]) / 1.5  # the procedural scene lives in radius 1.5: move the cameras with the box
data.all_c2w[:, :, 3] *= scale
BASE = (args.level_step // 16) * 16 if args.config == "neuralangelo" else 0
n_samples = 0
if args.modular:
    import refmirror
    import fixture_utils as fu
    model = refmirror.NeuSModel(cfg).to(dev).train()
    opt = torch.optim.AdamW(model.parameters(), lr=0.01, betas=(0.9, 0.99), eps=1e-15)
    gen = torch.Generator(device=dev).manual_seed(1)

    def step(i):
        global n_samples
        model.update_step(0, BASE + i)
        rays, rgb, fg, bg = data.sample_rays(args.rays, gen, cfg["background_color"])
        model.background_color = bg
        opt.zero_grad(set_to_none=True)
        out = model(rays)
        loss, _ = fu.neus_system_loss(out, rgb, fg, LAM)
        loss.backward()
        opt.step()
        n_samples += int(out["num_samples_full"].sum())
else:
    from nsr.fused_neus import NeuSTrainer
    model = nsr.build(cfg).to(dev).train()
    cfg_run = dict(cfg, train_num_rays=args.rays, dynamic_ray_sampling=args.dynamic)
    tr = NeuSTrainer(model, data, cfg_run, LAM, config_name=args.config)
    tr.global_step = BASE

    def step(i):
        global n_samples
        last = tr.train_step()
        n_samples += last["n_samples"] + last["n_samples_bg"]

for i in range(args.warmup):
    step(i)
torch.cuda.synchronize(); n_samples = 0; t0 = time.perf_counter()
for i in range(args.warmup, args.warmup + args.steps):
    step(i)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(json.dumps({"config": args.config, "path": "modular drop-in (autograd)" if args.modular else "fused (nsr/fused_neus.py)",
                  "rays_per_step": args.rays, "steps": args.steps, "ms_per_step": 1e3 * dt / args.steps,
                  "samples_per_step": n_samples / args.steps, "samples_per_sec": n_samples / dt,
                  "grad_type": cfg["geometry"]["grad_type"]}))
