"""Timing of one NeuS / neuralangelo training step (march, hash encode, fp32 SDF network with analytic normals or the
7-point finite-difference stencil, colour network, SDF->alpha compositing, system losses, backward, AdamW):
the FUSED runner (nsr/fused_neus.py) and, for comparison, the modular path through the drop-in packages (autograd over
~250 launches, the way the reference's models/neus.py drives them).  One JSON line.

    python tools/neus_step_bench.py [--config neus-blender|neuralangelo] [--rays 4096] [--steps 30] [--modular]
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
args = ap.parse_args()
torch.manual_seed(0)
dev = "cuda"
cfg = nsr.configs.get(args.config)
LAM = {"lambda_rgb_l1": 1.0, "lambda_mask": 0.1, "lambda_eikonal": 0.1}
data = SyntheticBlender(n_images=20, w=400, h=400, device=dev, seed=0)
gen = torch.Generator(device=dev).manual_seed(1)
if args.modular:
    import refmirror
    import fixture_utils as fu
    model = refmirror.NeuSModel(cfg).to(dev).train()
else:
    from nsr.fused_neus import FusedNeuSStep
    model = nsr.build(cfg).to(dev).train()
    fused = FusedNeuSStep(model, LAM)
opt = torch.optim.AdamW(model.parameters(), lr=0.01, betas=(0.9, 0.99), eps=1e-15)
scale = float(cfg["radius"]) / 1.5  # the procedural scene lives in radius 1.5
n_samples = 0
BASE = 0


def occ_refresh(i):
    """models/neus.py:90-111 through the drop-in occupancy grid (torch formulation) for either holder"""
    if args.modular:
        model.update_step(0, i)
        return
    model.update_step(0, i)
    if i % 16 == 0:
        import refmirror  # noqa: F401  (only for the closed-form alpha of the occupancy statistic)
        from nsr_hip import ops
        enc = fused.enc

        def occ_eval_fn(x):
            x01 = ops.contract_to_unisphere(x.float().contiguous(), fused.radius, 0)
            e = ops.hashgrid_forward(x01, enc.table_half(enc.params), enc.grid_desc, fused._mask_count())
            inp = torch.cat([x01 * 2 - 1, e.float()], -1)
            l0, l2 = fused.sdf.layers
            from nsr.fused_neus import _linear_weight
            z = torch.nn.functional.softplus(inp @ _linear_weight(l0).t() + l0.bias, beta=100)
            sdf = (z @ _linear_weight(l2).t() + l2.bias)[:, :1]
            inv_s = fused._inv_s().clip(1e-6, 1e6)
            h = model.render_step_size * 0.5
            p, nx = torch.sigmoid((sdf + h) * inv_s), torch.sigmoid((sdf - h) * inv_s)
            return ((p - nx + 1e-5) / (p + 1e-5)).clip(0.0, 1.0)
        with torch.no_grad():
            model.occupancy_grid.every_n_step(step=i, occ_eval_fn=occ_eval_fn,
                                              occ_thre=cfg.get("grid_prune_occ_thre", 0.01))


def step(i):
    global n_samples
    occ_refresh(BASE + i)
    rays, rgb, fg, bg = data.sample_rays(args.rays, gen, cfg["background_color"])
    rays = torch.cat([rays[:, :3] * scale, rays[:, 3:]], -1)
    model.background_color = bg
    opt.zero_grad(set_to_none=True)
    if args.modular:
        out = model(rays)
        loss, _ = fu.neus_system_loss(out, rgb, fg, LAM)
        loss.backward()
        n = int(out["num_samples"].sum())
    else:
        res = fused.forward_backward(rays, rgb, fg, bg)
        n = res["num_samples"]
        loss = res["loss_acc"][0]
    opt.step()
    n_samples += n
    return loss


# neuralangelo: run at the progressive level of --level-step (the schedule moves one level per 1000 steps)
BASE = (args.level_step // 16) * 16 if args.config == "neuralangelo" else 0
for i in range(args.warmup):
    step(i)
torch.cuda.synchronize(); n_samples = 0; t0 = time.perf_counter()
for i in range(args.warmup, args.warmup + args.steps):
    last = step(i)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(json.dumps({"config": args.config, "path": "modular drop-in (autograd)" if args.modular else "fused (nsr/fused_neus.py)",
                  "rays_per_step": args.rays, "steps": args.steps, "ms_per_step": 1e3 * dt / args.steps,
                  "samples_per_step": n_samples / args.steps, "samples_per_sec": n_samples / dt,
                  "grad_type": cfg["geometry"]["grad_type"]}))
