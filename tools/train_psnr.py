"""End-to-end quality check: train the nerf-blender config for N steps on the procedural scene through one of the three
tiers of the path, then render held-out views with the eval path (chunk_batch, models/nerf.py:111-127 + systems/nerf.py:118-160
semantics: white background, PSNR on the masked-composited image) and print one JSON line.

    --path fused     the asynchronous fused trainer (nsr.trainer.Trainer: what bench.py times)
    --path boundary  nsr.models.FusedNeRFModel behind the reference's model interface, the system's own statements around it
                     (systems/nerf.py:33-106: torch ray sampling, .item() on num_samples, smooth-L1 on boolean-masked rays,
                     loss.backward(), torch.optim.AdamW + MultiStepLR)
    --path modular   the reference's model statements (tests/refmirror = models/nerf.py:61-127) on the drop-in tinycudann /
                     nerfacc packages through autograd, Lightning's precision-16 protocol: torch.autocast(float16) +
                     GradScaler(65536) + torch.optim.AdamW + MultiStepLR -- the reference-semantics path
Same scene, same seed, same schedule, same evaluation for all three: BASELINE.json's "PSNR within 0.1 dB of reference" leg.

    python tools/train_psnr.py [--path fused] [--steps 20000] [--test-views 4] [--res 400]
"""
import argparse, json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "instant-nsr-pl_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import nsr
from nsr.scene import SyntheticBlender, get_rays

ap = argparse.ArgumentParser()
ap.add_argument("--path", default="fused", choices=["fused", "boundary", "modular"])
ap.add_argument("--steps", type=int, default=20000)
ap.add_argument("--test-views", type=int, default=4)
ap.add_argument("--res", type=int, default=400)
ap.add_argument("--seed", type=int, default=42)
args = ap.parse_args()
torch.manual_seed(args.seed)
dev = torch.device("cuda", 0)
# world > 1 (python -m torch.distributed.run --nproc-per-node 2 ... tools/train_psnr.py --path fused): ray-sharded data
# parallel through the trainer's own exchange; on a box with fewer GPUs than ranks the ranks share cuda:0 over gloo (RCCL
# refuses two ranks per device).  NSR_TRANSPORT=fp32|bf16 picks the table gradient's wire format.
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
extra = {}
if world > 1:
    import torch.distributed as dist
    assert args.path == "fused", "the multi-rank run goes through nsr.trainer.Trainer"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo" if torch.cuda.device_count() < world else "nccl")
elif os.environ.get("NSR_FORCE_SHARDED"):
    # ONE rank in an nccl (= RCCL) group: the trainer's multi-GPU exchange with its real wire format (NSR_TRANSPORT=bf16|fp32: the
    # table gradient is rounded to the wire format before the one-rank reduce-scatter), sharded AdamW and fp16 all-gather over
    # the WHOLE schedule -- what a one-GPU box can say about the bf16 default (the sum over ranks is not exercised)
    import torch.distributed as dist
    assert args.path == "fused"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    extra["exchange"] = {"backend": "nccl", "world": 1, "transport": os.environ.get("NSR_TRANSPORT", "bf16")}
cfg = nsr.configs.get("nerf-blender")
train = SyntheticBlender(n_images=100, w=args.res, h=args.res, device=dev, seed=0)
test = SyntheticBlender(n_images=args.test_views, w=args.res, h=args.res, device=dev, seed=12345)  # unseen cameras
milestones = [10000, 15000, 18000]

if args.path == "fused":
    from nsr.trainer import Trainer
    model = nsr.build(cfg).to(dev).train()
    tr = Trainer(model, train, cfg, rank=rank, world_size=world, seed=args.seed, async_mode=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.steps):
        tr.train_step()
    c = tr.counters(); dt = time.perf_counter() - t0
    final_loss = float(tr.last["loss"])
    n_samples, n_rays = c["samples"], c["rays"]
    extra["truncated_launches"] = c["truncated"]
    from nsr.export import render_rays
    render = lambda rays: render_rays(tr.fused, rays)["comp_rgb"]  # noqa: E731  (eval-mode chunked render, results on the CPU)
else:
    if args.path == "boundary":
        import nsr.models
        model = nsr.models.FusedNeRFModel(cfg).to(dev).train()
        opt = torch.optim.AdamW(model.parameters(), lr=0.01, betas=(0.9, 0.99), eps=1e-15, fused=True)
        scaler = None
    else:
        import refmirror
        model = refmirror.NeRFModel(cfg).to(dev).train()
        opt = torch.optim.AdamW(model.parameters(), lr=0.01, betas=(0.9, 0.99), eps=1e-15)
        scaler = torch.amp.GradScaler("cuda", init_scale=65536.0)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=milestones, gamma=0.33)
    gen = torch.Generator(device=dev); gen.manual_seed(args.seed)
    train_num_rays = cfg["train_num_rays"]
    target = cfg["train_num_rays"] * cfg["num_samples_per_ray"]
    n_samples = n_rays = skipped = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for step in range(args.steps):
        rays, rgb, fg, bg = train.sample_rays(train_num_rays, gen, cfg["background_color"])   # preprocess_data
        model.background_color = bg
        model.update_step(0, step)                                                            # on_train_batch_start
        with torch.autocast("cuda", dtype=torch.float16, enabled=scaler is not None):
            out = model(rays)                                                                  # training_step ...
            n = int(out["num_samples"].sum().item())
            if cfg["dynamic_ray_sampling"] and n > 0:
                t = int(train_num_rays * (target / n))
                train_num_rays = min(int(train_num_rays * 0.9 + t * 0.1), cfg["max_train_num_rays"])
            valid = out["rays_valid"][..., 0]
            loss = torch.nn.functional.smooth_l1_loss(out["comp_rgb"][valid], rgb[valid])
        opt.zero_grad(set_to_none=True)
        if scaler is not None:
            scaler.scale(loss).backward()
            before = scaler.get_scale()
            scaler.step(opt)
            scaler.update()
            skipped += int(scaler.get_scale() < before)
        else:
            loss.backward()
            opt.step()
        sched.step()
        n_samples += n
        n_rays += rays.shape[0]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    final_loss = float(loss.detach())
    if scaler is not None:
        extra.update(steps_skipped_by_grad_scaler=skipped, final_grad_scale=scaler.get_scale())
    model.eval()
    model.background_color = torch.ones(3, device=dev)

    def render(rays):
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16, enabled=scaler is not None):
            return model(rays)["comp_rgb"].float()

params_finite = all(bool(torch.isfinite(p).all()) for p in model.parameters())
model.eval()
psnrs = []
with torch.no_grad():
    for i in range(args.test_views):
        o, d = get_rays(test.directions.view(-1, 3), test.all_c2w[i:i + 1].expand(args.res * args.res, -1, -1))
        rays = torch.cat([o, torch.nn.functional.normalize(d, p=2, dim=-1)], -1)
        comp = render(rays)
        fg = test.all_fg_masks[i].view(-1, 1)
        gt = test.all_images[i].view(-1, 3) * fg + (1 - fg)
        mse = torch.mean((comp.to(dev).clamp(0, 1) - gt) ** 2)  # chunk_batch offloads to the CPU like the reference
        psnrs.append(float(-10.0 * torch.log10(mse)))
if world > 1:
    extra.update(world_size=world, transport=os.environ.get("NSR_TRANSPORT", "bf16"),
                 note="samples / rays per second are rank 0's own; every rank draws its own rays")
    dist.barrier()
if rank == 0:
  print(json.dumps(dict({"path": args.path, "steps": args.steps, "seed": args.seed, "train_seconds": dt,
                       "ms_per_step": 1e3 * dt / args.steps, "samples_per_sec": n_samples / dt, "rays_per_sec": n_rays / dt,
                       "final_train_loss": final_loss, "parameters_finite": params_finite,
                       "test_psnr": sum(psnrs) / len(psnrs), "test_psnr_per_view": psnrs, "test_res": args.res}, **extra)))
