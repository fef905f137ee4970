"""worker of tests/test_gpu_entry_protocol.py: two ranks (gloo rendezvous, both on the box's one GPU) drive the FUSED registry
entries the way the reference runs its models -- Lightning ``precision: 16`` (``torch.autocast(float16)`` +
``GradScaler(65536)``, configs/nerf-blender.yaml:103) under ``DistributedDataParallel(find_unused_parameters=False)``
(launch.py:93-107) -- with the statements of systems/nerf.py:87-106 / systems/neus.py:88-139 around ``model(rays)``.
Every rank also forms, on an unwrapped fp32 copy of the model, the gradients of BOTH ranks' batches; their mean is what DDP's
all-reduce + the scaler's unscale must leave in ``.grad``."""
import copy, json, os, sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "instant-nsr-pl_amd"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def rays_of(rank, n=600):
    g = torch.Generator().manual_seed(100 + rank)
    o = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1) * 4.0
    d = torch.nn.functional.normalize(-o + torch.randn(n, 3, generator=g) * 0.4, dim=-1)
    return torch.cat([o, d], -1).cuda(), torch.rand(n, 3, generator=g).cuda(), (torch.rand(n, generator=g) > 0.3).float().cuda()


def nerf_loss(model, batch):
    rays, rgb, _ = batch
    out = model(rays)
    n = int(out["num_samples"].sum().item())
    valid = out["rays_valid"][..., 0]
    return torch.nn.functional.smooth_l1_loss(out["comp_rgb"][valid], rgb[valid]), n


def neus_loss(model, batch):
    rays, rgb, fg = batch
    F = torch.nn.functional
    out = model(rays)
    n = int(out["num_samples_full"].sum().item())
    valid = out["rays_valid_full"][..., 0]
    loss = 10.0 * F.mse_loss(out["comp_rgb_full"][valid], rgb[valid])
    loss = loss + 0.1 * ((torch.linalg.norm(out["sdf_grad_samples"], ord=2, dim=-1) - 1.0) ** 2).mean()
    opacity = torch.clamp(out["opacity"].squeeze(-1), 1.0e-3, 1.0 - 1.0e-3)
    # systems/criterions.py:155-159: the reference's own binary_cross_entropy (F.binary_cross_entropy refuses autocast)
    return loss + 0.1 * (-(fg * torch.log(opacity) + (1 - fg) * torch.log(1 - opacity)).mean()), n


def build(kind):
    import nsr, nsr.models
    torch.manual_seed(7)
    if kind == "nerf":
        cfg = nsr.configs.get("nerf-blender")
        m = nsr.models.FusedNeRFModel(cfg).cuda().train()
        with torch.no_grad():
            m.geometry.encoding_with_network.params[3072:].normal_(0, 0.08)
        radius = 1.1
    else:
        cfg = nsr.configs.get("neus-blender")
        m = nsr.models.FusedNeuSModel(cfg).cuda().train()
        with torch.no_grad():  # sphere init zeroes the first layer's encoding columns: the table would see no gradient at all
            w = m.geometry.network.layers.get_submodule("0").weight_v
            w[:, 3:] += 0.05 * torch.randn(w[:, 3:].shape, device=w.device, generator=torch.Generator(device=w.device).manual_seed(3))
            m.geometry.encoding.encoding.params.normal_(0, 0.02, generator=torch.Generator(device=w.device).manual_seed(4))
        radius = 0.9
    ii = torch.stack(torch.meshgrid(*[torch.arange(128)] * 3, indexing="ij"), -1).float().cuda()
    r = float(cfg["radius"])
    m.occupancy_grid._binary = (((ii + 0.5) / 128 * 2 * r - r).norm(dim=-1) < radius)
    m.randomized = False
    m.background_color = torch.tensor([0.3, 0.6, 0.9], device="cuda")
    return m


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    report = {}
    for kind, loss_fn in (("nerf", nerf_loss), ("neus", neus_loss)):
        model = build(kind)
        plain = build(kind)  # fp32, no autocast, no scaler, no DDP: the reference gradients
        plain.load_state_dict(model.state_dict())
        want = None
        for r in range(world):
            plain.zero_grad(set_to_none=True)
            loss, _ = loss_fn(plain, rays_of(r))
            loss.backward()
            gs = {k: p.grad.detach().clone() for k, p in plain.named_parameters() if p.grad is not None}
            want = gs if want is None else {k: want[k] + gs[k] for k in want}
        want = {k: v / world for k, v in want.items()}
        ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], find_unused_parameters=False)
        params = [p for p in model.parameters() if p.numel() > 0]
        opt = torch.optim.AdamW(params, lr=0.01, betas=(0.9, 0.99), eps=1e-15)
        scaler = torch.amp.GradScaler("cuda", init_scale=65536.0)
        with torch.autocast("cuda", dtype=torch.float16):
            loss, n = loss_fn(ddp, rays_of(rank))
        opt.zero_grad(set_to_none=True)
        scaler.scale(loss).backward()
        scaler.unscale_(opt)
        got = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
        errs = {k: rel(got[k], want[k]) for k in want if want[k].numel() > 0 and float(want[k].abs().max()) > 0}
        before = {k: p.detach().clone() for k, p in model.named_parameters()}
        scaler.step(opt)
        scaler.update()
        moved = sum(int(not torch.equal(before[k], p.detach())) for k, p in model.named_parameters() if p.numel() > 0)
        # replicas identical after the step
        flat = torch.cat([p.detach().reshape(-1).float() for p in params])
        other = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(other, flat)
        mismatch = max(float((o - flat).abs().max()) for o in other)
        # an overflowing scale: inf / NaN must reach .grad, the scaler skips the step and backs off
        scaler2 = torch.amp.GradScaler("cuda", init_scale=2.0 ** 127)
        with torch.autocast("cuda", dtype=torch.float16):
            loss2, _ = loss_fn(ddp, rays_of(rank))
        opt.zero_grad(set_to_none=True)
        scaler2.scale(loss2).backward()
        keep = {k: p.detach().clone() for k, p in model.named_parameters()}
        scaler2.step(opt)
        scaler2.update()
        unchanged = all(torch.equal(keep[k], p.detach()) for k, p in model.named_parameters())
        report[kind] = {"samples": n, "grad_norms": {k: float(v.double().norm()) for k, v in want.items()}, "missing_grads": sorted(set(want) - set(got)), "max_rel_err": max(errs.values()),
                        "worst": max(errs, key=errs.get), "errs": {k: round(v, 6) for k, v in errs.items()},
                        "tensors_moved": moved, "tensors": len(params), "replica_mismatch": mismatch,
                        "overflow_step_skipped": bool(unchanged), "scale_after_overflow": scaler2.get_scale(),
                        "finite": bool(all(torch.isfinite(p).all() for p in params))}
        del ddp, model, plain
        torch.cuda.empty_cache()
    if rank == 0:
        print("ENTRY_PROTOCOL_REPORT " + json.dumps(report), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
