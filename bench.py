#!/usr/bin/env python
"""bench.py -- the hot path of bennyguo/instant-nsr-pl on MI355X: one training step of configs/nerf-blender.yaml
(HashGrid L16 T2^19 F2 + fused 64-wide MLPs, <= 8192 rays/step) = sample rays -> occupancy refresh -> march ->
hash-encode + MLP -> composite -> loss -> backward -> (RCCL grad all-reduce) -> AdamW.   Synthetic scene.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Either launched by ``python -m torch.distributed.run --nproc-per-node N ... bench.py
--gpus N`` (RANK / WORLD_SIZE in the environment) or, from a bare ``python bench.py --gpus N``, this script re-launches
itself that way (the reference's launch.py:93-107 spawns its own ranks too).  With fewer GPUs than ranks (a 1-GPU
development box) the ranks share device 0 over a gloo rendezvous: a smoke run of the multi-rank code path, not a measurement.

Prints ONE JSON line (rank 0).  value = live hash-encoded MLP samples/s over the whole job (all ranks).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "instant-nsr-pl_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0           # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ENC_FWD_BYTES_PER_SAMPLE = 588  # 12 B coords + 16*8*2*2 B gathers + 64 B out (fp16 table)
ADAM_TABLE_BYTES_PER_PARAM = 26  # fp32 p, m, v read (12) + written (12) + fp16 image written (2)
# table backward with AdamW applied inside it: the table gradient never reaches HBM (it lives in the LDS of the workgroup
# that owns the slice), so SURVEY 8(d)'s 2 x 2048 B/sample of gradient read-modify-write do not exist for this launch --
# what it must move is each sample's position (12 B) + level-major dy (16 levels x 2 x 4 B) and the optimizer's bytes
ENC_BWD_FUSED_INPUT_BYTES_PER_SAMPLE = 12 + 16 * 2 * 4
ENC_BWD_BYTES_PER_SAMPLE = 2124  # 12 + 64*2(fp32 dy) ... fp32 atomics: 16*8*2*4 B *2 (RMW) + coords + dy


def _time_loop(step, seconds_budget, max_reps):
    step()  # warm-up
    t0, reps = time.time(), 0
    while True:
        step()
        reps += 1
        if time.time() - t0 > seconds_budget or reps >= max_reps:
            break
    return reps, time.time() - t0


def cpu_baseline(seconds_budget=8.0):
    """The reference's CPU formulation of the path, timed on this box's host cores (fp32, all cores, N = 2^18 = one
    training step's sample budget, forward + backward):
      (a) "reference": VanillaFrequency(10) + VanillaMLP density head + SH + VanillaMLP colour head -- the reference's own
          pure-PyTorch encoding + MLP (models/network_utils.py:14-37,95-139; BASELINE.json configs[0]), restated in
          oracle/vanilla_ref.py and pinned against the reference modules by tests/test_oracle_vanilla.py;
      (b) "hashgrid_port": the oracle's pure-PyTorch hash grid + fused-MLP restatement (the reference has no CPU hash grid of
          its own) on the same inputs.
    value = (a); both are reported.  kind = "port" (restatements: /root/reference does not exist on the GPU box)."""
    from oracle import tcnn_ref, vanilla_ref
    import nsr
    cfg = nsr.configs.get("nerf-blender")
    cores = int(os.environ.get("NSR_CPU_BASELINE_THREADS", min(os.cpu_count(), 32)))  # measured on the 256-thread GPU box
    torch.set_num_threads(cores)  # (profiles/r02_cpu_baseline_thread_sweep.json): 16-32 threads 9.6e5 samples/s, 64: 4.9e5, 256: 1e4
    n = 1 << 18
    g = torch.Generator().manual_seed(0)
    x = torch.rand(n, 3, generator=g)
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    sh = tcnn_ref.Encoding(3, cfg["texture"]["dir_encoding_config"])
    # (a) the reference's CPU encoding + MLP
    freq = vanilla_ref.VanillaFrequency(3, {"n_frequencies": 10})
    dens = vanilla_ref.VanillaMLP(3 + freq.n_output_dims, 16, 64, 1)
    col = vanilla_ref.VanillaMLP(16 + 16, 3, 64, 2)

    def step_a():
        feat = dens(vanilla_ref.include_xyz(freq, x))
        rgb = col(torch.cat([feat, sh((d + 1) / 2).float()], -1))
        (rgb.sum() + feat[:, 0].sum()).backward()

    reps_a, dt_a = _time_loop(step_a, seconds_budget, 50)
    # (b) the hash-grid port
    torch.set_num_threads(min(cores, 32))  # the dense index_add backward stops scaling beyond a few dozen threads
    ewn = tcnn_ref.NetworkWithInputEncoding(3, 16, cfg["geometry"]["xyz_encoding_config"],
                                            cfg["geometry"]["mlp_network_config"])
    net = tcnn_ref.Network(32, 3, cfg["texture"]["mlp_network_config"])
    nb = 1 << 16
    xb, db = x[:nb], d[:nb]

    def step_b():
        feat = ewn(xb).float()
        rgb = net(torch.cat([feat, sh((db + 1) / 2).float()], -1)).float()
        (rgb.sum() + feat[:, 0].sum()).backward()

    reps_b, dt_b = _time_loop(step_b, seconds_budget, 20)
    b3, b4 = cpu_marcher_compositor(cores), cpu_config0_step(cores)
    return {"marcher_compositor": b3, "config0_end_to_end": b4, "value": n * reps_a / dt_a, "unit": "samples/s", "cores": cores, "host_logical_cpus": os.cpu_count(), "kind": "port",
            "sample": f"{reps_a} x 2^18 uniform samples, VanillaFrequency(10)+xyz -> VanillaMLP(64x1) -> 16, SH4, "
                      f"VanillaMLP(64x2) -> rgb, forward + backward, fp32 torch on {cores} host threads "
                      f"(oracle/vanilla_ref.py = models/network_utils.py:14-37,95-139)",
            "hashgrid_port": {"value": nb * reps_b / dt_b, "unit": "samples/s", "cores": torch.get_num_threads(),
                              "sample": f"{reps_b} x 2^16 uniform samples, hash-encode (L16 T2^19 F2) + fused-MLP "
                                        f"restatement (oracle/tcnn_ref.py), forward + backward, fp32"}}


def _cpu_rays(n_rays, seed=0):
    """camera rays of the Blender setup (origins on the r = 4.03 sphere, looking at the scene of radius 1.5)"""
    g = torch.Generator().manual_seed(seed)
    o = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=g), dim=-1) * 4.03
    target = (torch.rand(n_rays, 3, generator=g) * 2 - 1) * 0.8
    return o, torch.nn.functional.normalize(target - o, dim=-1)


def _cpu_ball_grid():
    from oracle import nerfacc_ref
    grid = nerfacc_ref.OccupancyGrid(torch.tensor([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5]), 128)
    c = (torch.arange(128) + 0.5) / 128 * 3.0 - 1.5
    zz, yy, xx = torch.meshgrid(c, c, c, indexing="ij")
    grid._binary[:] = ((xx * xx + yy * yy + zz * zz) < 1.0).reshape(grid._binary.shape)  # a ball: ~15 % of the cells
    return grid


def cpu_marcher_compositor(cores, seconds_budget=3.0):
    """BASELINE.md section 3, leg B3: the CPU oracle's ray marcher (scalar C, oracle/csrc/nerfacc_ref.c, 1 thread) +
    render_weight_from_density + 3 x accumulate_along_rays (torch, `cores` threads) at 1,024 and 8,192 rays through a
    128^3 occupancy grid, step 0.00507 (nerfacc==0.3.3 call sites models/nerf.py:83,105-108)"""
    from oracle import nerfacc_ref as N
    torch.set_num_threads(cores)
    grid = _cpu_ball_grid()
    aabb = torch.tensor([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5])
    out = {}
    for n_rays in (1024, 8192):
        o, d = _cpu_rays(n_rays)
        box = {}

        def step():
            ri, t0, t1 = N.ray_marching(o, d, scene_aabb=aabb, grid=grid, render_step_size=0.00507421875)
            sig = torch.full_like(t0, 20.0)
            w = N.render_weight_from_density(t0, t1, sig, ray_indices=ri, n_rays=n_rays)
            N.accumulate_along_rays(w, ri, values=None, n_rays=n_rays)
            N.accumulate_along_rays(w, ri, values=(t0 + t1) / 2, n_rays=n_rays)
            N.accumulate_along_rays(w, ri, values=torch.ones(len(ri), 3), n_rays=n_rays)
            box["n"] = len(ri)

        reps, dt = _time_loop(step, seconds_budget, 200)
        out[str(n_rays)] = {"rays_per_sec": n_rays * reps / dt, "samples_per_sec": box["n"] * reps / dt,
                            "samples_per_call": box["n"], "calls": reps}
    out["what"] = "oracle marcher (C, 1 thread) + render_weight_from_density + 3 accumulate_along_rays (torch), forward"
    return out


def cpu_config0_step(cores, seconds_budget=5.0):
    """BASELINE.md section 3, leg B4 = BASELINE.json configs[0]: one training step (march with sigma_fn pruning -> fields
    -> composite -> smooth-L1 -> backward -> AdamW) of the NeRF model at 1,024 rays with the reference's pure-PyTorch
    encoding + MLPs (VanillaFrequency + VanillaMLP, oracle/vanilla_ref.py) on the CPU oracle backend
    (oracle/glue_ref.py:nerf_forward = models/nerf.py:61-127)"""
    from oracle import glue_ref, tcnn_ref, vanilla_ref
    import nsr
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    cfg = nsr.configs.get("nerf-blender")
    grid = _cpu_ball_grid()
    aabb = torch.tensor([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5])
    freq = vanilla_ref.VanillaFrequency(3, {"n_frequencies": 10})
    dens = vanilla_ref.VanillaMLP(3 + freq.n_output_dims, 16, 64, 1)
    col = vanilla_ref.VanillaMLP(16 + 16, 3, 64, 2)
    sh = tcnn_ref.Encoding(3, cfg["texture"]["dir_encoding_config"])
    opt = torch.optim.AdamW(list(dens.parameters()) + list(col.parameters()), lr=0.01, betas=(0.9, 0.99), eps=1e-15)
    o, d = _cpu_rays(1024)
    rays, gt = torch.cat([o, d], -1), torch.rand(1024, 3)
    box = {}

    def step():
        out = glue_ref.nerf_forward(rays, lambda x: dens(vanilla_ref.include_xyz(freq, x)), lambda u: sh(u).float(), col,
                                    grid, aabb, 1.5, 0.00507421875, torch.ones(3))
        valid = out["rays_valid"][..., 0]
        loss = torch.nn.functional.smooth_l1_loss(out["comp_rgb"][valid], gt[valid])
        opt.zero_grad()
        loss.backward()
        opt.step()
        box["n"] = int(out["num_samples"])

    reps, dt = _time_loop(step, seconds_budget, 50)
    return {"rays_per_sec": 1024 * reps / dt, "samples_per_sec": box["n"] * reps / dt, "ms_per_step": 1e3 * dt / reps,
            "kept_samples_per_step": box["n"], "steps": reps,
            "what": "1,024 rays/step, VanillaFrequency(10)+xyz -> VanillaMLP density, SH4 -> VanillaMLP colour, CPU marcher + "
                    "compositor, smooth-L1, backward, AdamW"}


def pmc_traffic(name, samples_per_launch, warmup=None, steps=None):
    """HBM-side bytes per launch of the dominant operation from the PMC passes of THIS round (tools/collect_profiles.sh ->
    profiles/r06_pmc_traffic.json (older rounds' files behind it), assembled by tools/pmc_traffic.py: one entry per (warmup, steps) regime the passes were
    run in) -- used only when a pass ran in the same regime (same warmup / steps, or its recorded samples per launch
    within 15 % of this run's); otherwise the field is null"""
    path = next((q for q in (os.path.join(ROOT, "profiles", f) for f in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json",
                                                                         "r02_pmc_traffic.json")) if os.path.exists(q)), None)
    if path is None:
        return None, "no PMC file"
    pmc = json.load(open(path))
    regimes = pmc.get("regimes")
    if regimes:  # pick the pass that ran with this command line, else the closest one by samples per launch
        key = f"w{warmup}_s{steps}"
        cands = [regimes[key]] if key in regimes else []
        cands += sorted((r for k, r in regimes.items() if k != key and isinstance(r.get(name), dict)),
                        key=lambda r: abs(r[name].get("samples_per_launch", 0) - samples_per_launch))
        pmc = cands[0] if cands else {}
    ent = pmc.get(name)
    if not isinstance(ent, dict) or not ent.get("samples_per_launch"):
        return None, "PMC file has no entry / regime for " + name
    ratio = samples_per_launch / ent["samples_per_launch"]
    if not 0.85 <= ratio <= 1.15:
        return None, (f"PMC passes ran at {ent['samples_per_launch']:.0f} samples/launch, this run at "
                      f"{samples_per_launch:.0f}: not comparable")
    return ent["bytes_per_launch"], f"profiles/{os.path.basename(path)} ({ent['samples_per_launch']:.0f} samples/launch)"


def other_workloads(dev, rank=0, world=1, sync=None, n_steps=60):
    """the fused NeuS (C3), NeuS + NeRF++ background (C4) and neuralangelo (C5) steps at the reference's operating point (dynamic ray count targeting 2^18
    samples/step), 60 + 60 steps each: reported beside the headline line, never part of `value`.  world > 1: ray-sharded
    over all ranks (BASELINE.json names C4 / C5 as the 8-GPU configs), same barrier + max-over-ranks timing as the headline"""
    import nsr
    from nsr.fused_neus import NeuSTrainer
    from nsr.scene import SyntheticBlender
    out = {}
    lam = {"neus-blender": {"lambda_rgb_mse": 10.0, "lambda_rgb_l1": 0.0, "lambda_mask": 0.1, "lambda_eikonal": 0.1},
           "neus-dtu": {"lambda_rgb_mse": 0.0, "lambda_rgb_l1": 1.0, "lambda_mask": 0.0, "lambda_eikonal": 0.1},
           "neuralangelo": {"lambda_rgb_mse": 0.0, "lambda_rgb_l1": 1.0, "lambda_mask": 0.1, "lambda_eikonal": 0.1}}
    for name in ("neus-blender", "neus-dtu", "neuralangelo"):
        try:
            torch.manual_seed(7)
            cfg = nsr.configs.get(name)
            data = SyntheticBlender(n_images=20, w=400, h=400, device=dev, seed=0,
                                    environment=bool(cfg["learned_background"]))
            data.all_c2w[:, :, 3] *= float(cfg["radius"]) / 1.5
            model = nsr.build(cfg).to(dev).train()
            tr = NeuSTrainer(model, data, cfg, lam[name], config_name=name, rank=rank, world_size=world)
            if name == "neuralangelo":
                tr.global_step = 12000  # all 16 levels active (the schedule moves one level per 1000 steps)
            for _ in range(n_steps):
                tr.train_step()
            (sync or torch.cuda.synchronize)()
            t0, n = time.perf_counter(), 0
            for _ in range(n_steps):
                last = tr.train_step()
                n += last["n_samples"] + last["n_samples_bg"]  # num_samples_full: foreground + NeRF++ background
            (sync or torch.cuda.synchronize)()
            dt = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([dt, float(n)], dtype=torch.float64, device=dev)
                tm = t[:1].clone()
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                dist.all_reduce(t[1:], op=dist.ReduceOp.SUM)
                dt, n = float(tm[0]), float(t[1])
            out[name] = {"ms_per_step": 1e3 * dt / n_steps, "samples_per_sec": n / dt, "samples_per_step": n / n_steps,
                         "n_gpus": world, "grad_type": cfg["geometry"]["grad_type"], "path": "nsr.fused_neus.NeuSTrainer"}
            del tr, model, data
            torch.cuda.empty_cache()
        except Exception as e:  # noqa: BLE001  (never let a side measurement take the headline line down)
            if world > 1:
                raise  # ... except in a multi-rank run, where a rank that skips ahead would desynchronise the collectives
            out[name] = {"error": repr(e)[:300]}
    return out


def _median_pass(fn, passes):
    """``passes`` runs of a host-bound side measurement back to back; the MEDIAN pass (by ms per step) with every pass's figure:
    one pass of these python-driven loops moves by 30-50 % from run to run on the same code (profiles/
    r06_modular_path_r4_vs_now.json; boundary_path in round 6: 1.29 / 1.64 / 1.98 ms in one process)"""
    runs = [fn() for _ in range(max(1, passes))]
    order = sorted(range(len(runs)), key=lambda k: runs[k]["ms_per_step"])
    res = dict(runs[order[len(order) // 2]])
    res["passes_ms_per_step"] = [round(r["ms_per_step"], 3) for r in runs]
    res["reported"] = "median pass"
    return res


def boundary_path(dev, passes=3):
    """three passes of ``boundary_pass``, the median reported (see _median_pass)"""
    return _median_pass(lambda: boundary_pass(dev), passes)


def boundary_path_neus(dev, passes=3):
    """three passes of ``boundary_pass_neus``, the median reported (see _median_pass)"""
    return _median_pass(lambda: boundary_pass_neus(dev), passes)


def boundary_pass(dev, warmup=300, steps=200, late_at=3000, late_steps=200):
    """The same training step driven THROUGH THE DROP-IN BOUNDARY the way the reference's system drives its model
    (systems/nerf.py:33-99, systems/base.py:54-57): torch ray sampling -> model.update_step -> out = model(rays) ->
    dynamic ray count from out['num_samples'] -> smooth-L1 on the valid rays in torch -> loss.backward() ->
    torch.optim.AdamW + MultiStepLR.  The model is nsr.models.FusedNeRFModel = what `models.make('nerf', cfg)` builds once the
    registry is pointed at it: forward and backward are one autograd.Function over the fused C passes.
    Reported beside the headline (whose trainer also owns loss / optimizer / ray sampling); never part of `value`."""
    import nsr
    import nsr.models
    from nsr.scene import SyntheticBlender
    torch.manual_seed(42)
    cfg = nsr.configs.get("nerf-blender")
    model = nsr.models.FusedNeRFModel(cfg).to(dev).train()
    data = SyntheticBlender(n_images=100, w=800, h=800, device=dev, seed=0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(42)
    try:
        opt = torch.optim.AdamW(model.parameters(), lr=0.01, betas=(0.9, 0.99), eps=1e-15, fused=True)
        opt_kind = "torch.optim.AdamW(fused=True)"
    except Exception:  # noqa: BLE001
        opt = torch.optim.AdamW(model.parameters(), lr=0.01, betas=(0.9, 0.99), eps=1e-15)
        opt_kind = "torch.optim.AdamW"
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[10000, 15000, 18000], gamma=0.33)
    train_num_rays = cfg["train_num_rays"]
    target = cfg["train_num_rays"] * cfg["num_samples_per_ray"]
    n_samples = n_rays = 0
    t0 = dt = None
    late = {"samples": 0, "rays": 0, "t0": None}
    for step in range(max(warmup + steps, late_at + late_steps if late_steps else 0)):
        if step == warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        if step == warmup + steps:
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            first_loss = float(loss.detach())
        if late_steps and step == late_at:
            torch.cuda.synchronize()
            late["t0"] = time.perf_counter()
        rays, rgb, fg, bg = data.sample_rays(train_num_rays, gen, cfg["background_color"])  # preprocess_data
        model.background_color = bg
        model.update_step(0, step)                                                           # on_train_batch_start
        out = model(rays)                                                                    # training_step ...
        n = int(out["num_samples"].sum().item())
        if cfg["dynamic_ray_sampling"] and n > 0:
            t = int(train_num_rays * (target / n))
            train_num_rays = min(int(train_num_rays * 0.9 + t * 0.1), cfg["max_train_num_rays"])
        valid = out["rays_valid"][..., 0]
        loss = torch.nn.functional.smooth_l1_loss(out["comp_rgb"][valid], rgb[valid])
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        sched.step()
        if warmup <= step < warmup + steps:
            n_samples += n
            n_rays += rays.shape[0]
        if late["t0"] is not None:
            late["samples"] += n
            late["rays"] += rays.shape[0]
    torch.cuda.synchronize()
    t_end = time.perf_counter()
    if dt is None:
        dt, first_loss = t_end - t0, float(loss.detach())
    res = {"samples_per_sec": n_samples / dt, "ms_per_step": 1e3 * dt / steps, "train_rays_per_sec": n_rays / dt,
           "kept_samples_per_step": n_samples / steps, "rays_per_step": n_rays / steps, "final_loss": first_loss,
           "warmup": warmup, "steps": steps, "optimizer": opt_kind, "model": "nsr.models.FusedNeRFModel",
           "lazy_outputs": bool(getattr(model, "lazy_outputs", False)),
           "what": "reference-style step through the model interface: torch ray sampling, model.update_step, model(rays), "
                   ".item() on num_samples, torch smooth-L1 on boolean-masked rays, loss.backward(), torch AdamW, MultiStepLR"}
    if late["t0"] is not None:
        dl = t_end - late["t0"]
        res["late"] = {"at_step": late_at, "timed_steps": late_steps, "ms_per_step": 1e3 * dl / late_steps,
                       "samples_per_sec": late["samples"] / dl, "kept_samples_per_step": late["samples"] / late_steps,
                       "rays_per_step": late["rays"] / late_steps, "final_loss": float(loss.detach()),
                       "note": "the same loop run on to a pruned grid: where the entry's removed host waits show (DESIGN 5.1b)"}
    return res


def modular_path(dev, warmup=60, steps=100, passes=3):
    """``passes`` runs of ``modular_pass`` back to back in this process; reports the MEDIAN pass and every pass's ms / step.
    One pass is a host-bound loop (~300 torch launches and ~11 host synchronisations per step) whose time moves 3.1 - 4.7 ms
    from one pass to the next on the same code (profiles/r06_modular_path_r4_vs_now.json: round 5's "3.31 -> 3.81 ms
    regression" was one pass against one pass inside that spread -- the round-4 tree re-measured beside this one gives
    3.4 - 4.1 ms)."""
    return _median_pass(lambda: modular_pass(dev, warmup, steps), passes)


def modular_pass(dev, warmup=60, steps=100):
    """The reference's OWN model code path on the drop-in packages: ``models/nerf.py:61-127`` statement by statement (its
    restatement in tests/refmirror -- /root/reference itself cannot travel to the GPU box) over ``tinycudann`` / ``nerfacc`` =
    the HIP packages, every call through autograd, driven the way Lightning drives it with ``precision: 16``
    (configs/nerf-blender.yaml:103): ``torch.autocast(float16)`` + ``GradScaler(init_scale=65536)`` + ``torch.optim.AdamW`` +
    MultiStepLR, the system's statements of systems/nerf.py:33-106 around it.  The slowest of the three tiers (fused trainer >
    fused model entry > this): what a maintainer gets by only putting ``instant-nsr-pl_amd`` on PYTHONPATH."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import nsr
    import refmirror
    from nsr.scene import SyntheticBlender
    torch.manual_seed(42)
    cfg = nsr.configs.get("nerf-blender")
    model = refmirror.NeRFModel(cfg).to(dev).train()
    data = SyntheticBlender(n_images=100, w=800, h=800, device=dev, seed=0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(42)
    opt = torch.optim.AdamW(model.parameters(), lr=0.01, betas=(0.9, 0.99), eps=1e-15)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[10000, 15000, 18000], gamma=0.33)
    scaler = torch.amp.GradScaler("cuda", init_scale=65536.0)
    train_num_rays = cfg["train_num_rays"]
    target = cfg["train_num_rays"] * cfg["num_samples_per_ray"]
    n_samples = n_rays = skipped = 0
    t0 = None
    for step in range(warmup + steps):
        if step == warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        rays, rgb, fg, bg = data.sample_rays(train_num_rays, gen, cfg["background_color"])
        model.background_color = bg
        model.update_step(0, step)
        with torch.autocast("cuda", dtype=torch.float16):
            out = model(rays)
            n = int(out["num_samples"].sum().item())
            if cfg["dynamic_ray_sampling"] and n > 0:
                t = int(train_num_rays * (target / n))
                train_num_rays = min(int(train_num_rays * 0.9 + t * 0.1), cfg["max_train_num_rays"])
            valid = out["rays_valid"][..., 0]
            loss = torch.nn.functional.smooth_l1_loss(out["comp_rgb"][valid], rgb[valid])
        opt.zero_grad(set_to_none=True)
        scaler.scale(loss).backward()
        scale_before = scaler.get_scale()
        scaler.step(opt)
        scaler.update()
        skipped += int(scaler.get_scale() < scale_before)
        sched.step()
        if step >= warmup:
            n_samples += n
            n_rays += rays.shape[0]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"samples_per_sec": n_samples / dt, "ms_per_step": 1e3 * dt / steps, "train_rays_per_sec": n_rays / dt,
            "kept_samples_per_step": n_samples / steps, "rays_per_step": n_rays / steps, "final_loss": float(loss.detach()),
            "warmup": warmup, "steps": steps, "steps_skipped_by_grad_scaler": skipped, "grad_scale": scaler.get_scale(),
            "optimizer": "torch.optim.AdamW + GradScaler(65536) under torch.autocast(float16)",
            "model": "tests/refmirror NeRFModel (= models/nerf.py:61-127) on the drop-in tinycudann / nerfacc packages",
            "what": "the reference's own model statements through autograd on the HIP packages, Lightning precision-16 protocol; "
                    "early in training (steps 60-160 of a fresh model: few rays, dense grid)"}


def boundary_pass_neus(dev, warmup=100, steps=100):
    """configs[2] (neus-blender) driven through the drop-in boundary the way the reference's NeuS system drives its model
    (systems/neus.py:88-139): torch ray sampling -> model.update_step -> out = model(rays) -> dynamic ray count from
    out['num_samples_full'] -> MSE / L1 on the valid rays, eikonal on sdf_grad_samples, mask BCE on opacity, in torch ->
    loss.backward() -> torch AdamW with the YAML's parameter groups.  Model: nsr.models.FusedNeuSModel."""
    import nsr
    import nsr.models
    from nsr.scene import SyntheticBlender
    torch.manual_seed(7)
    cfg = nsr.configs.get("neus-blender")
    model = nsr.models.FusedNeuSModel(cfg).to(dev).train()
    data = SyntheticBlender(n_images=20, w=400, h=400, device=dev, seed=0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    var = [model.variance.variance]
    rest = [p for p in model.parameters() if p is not var[0] and p.numel() > 0]
    groups = [{"params": rest, "lr": 0.01}, {"params": var, "lr": 0.001}]
    try:
        opt = torch.optim.AdamW(groups, betas=(0.9, 0.99), eps=1e-15, fused=True)
        opt_kind = "torch.optim.AdamW(fused=True)"
    except Exception:  # noqa: BLE001
        opt = torch.optim.AdamW(groups, betas=(0.9, 0.99), eps=1e-15)
        opt_kind = "torch.optim.AdamW"
    train_num_rays = cfg["train_num_rays"]
    target = cfg["train_num_rays"] * (cfg["num_samples_per_ray"] + cfg.get("num_samples_per_ray_bg", 0))
    n_samples = n_rays = 0
    t0 = None
    F = torch.nn.functional
    for step in range(warmup + steps):
        if step == warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        rays, rgb, fg, bg = data.sample_rays(train_num_rays, gen, cfg["background_color"])
        model.background_color = bg
        model.update_step(0, step)
        out = model(rays)
        n = int(out["num_samples_full"].sum().item())
        if cfg["dynamic_ray_sampling"] and n > 0:
            t = int(train_num_rays * (target / n))
            train_num_rays = min(int(train_num_rays * 0.9 + t * 0.1), cfg["max_train_num_rays"])
        valid = out["rays_valid_full"][..., 0]
        loss = 10.0 * F.mse_loss(out["comp_rgb_full"][valid], rgb[valid])
        loss = loss + 0.1 * ((torch.linalg.norm(out["sdf_grad_samples"], ord=2, dim=-1) - 1.0) ** 2).mean()
        opacity = torch.clamp(out["opacity"].squeeze(-1), 1.0e-3, 1.0 - 1.0e-3)
        loss = loss + 0.1 * F.binary_cross_entropy(opacity, fg.float())
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        if step >= warmup:
            n_samples += n
            n_rays += rays.shape[0]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"samples_per_sec": n_samples / dt, "ms_per_step": 1e3 * dt / steps, "samples_per_step": n_samples / steps,
            "rays_per_step": n_rays / steps, "final_loss": float(loss.detach()), "warmup": warmup, "steps": steps,
            "optimizer": opt_kind, "model": "nsr.models.FusedNeuSModel (neus-blender)",
            "what": "reference-style NeuS step through the model interface: torch ray sampling, model.update_step, model(rays), "
                    "torch MSE / eikonal / mask losses, loss.backward(), torch AdamW"}


def whole_run(dev, data, cfg, n_steps=20000, late_at=10000, late_steps=200, test_views=4, test_res=400, seed=43):
    """The WHOLE training run of the reference schedule (configs/nerf-blender.yaml:96: 20,000 steps under the 8,192-ray cap,
    MultiStepLR milestones 10k / 15k / 18k) with a fresh model through the same asynchronous trainer the driver-timed region
    uses: wall time from the first step to the end of the last (one synchronisation), kept samples / rays from the device
    counters, PSNR on unseen views through the eval path (PSNR@20k is part of BASELINE.json's metric); inside it, `late_steps`
    steps at step `late_at` are bracketed by two synchronisations -> the LATE regime (sparse grid, ~4e4 kept samples per step),
    which is where a run spends most of its steps.  -> (whole_run, late_regime)"""
    import nsr
    from nsr.export import render_rays
    from nsr.scene import SyntheticBlender, get_rays
    from nsr.trainer import Trainer
    torch.manual_seed(seed)
    model = nsr.build(cfg).to(dev).train()
    tr = Trainer(model, data, cfg, rank=0, world_size=1, seed=seed, async_mode=True)
    late = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = 0
    while k < n_steps:
        if k == late_at and late_at + late_steps <= n_steps:
            torch.cuda.synchronize()
            c0, tl = tr.counters(), time.perf_counter()
            for _ in range(late_steps):
                tr.train_step()
            th = time.perf_counter()
            torch.cuda.synchronize()
            dtl = time.perf_counter() - tl
            c1 = tr.counters()
            late = {"at_step": late_at, "timed_steps": late_steps, "ms_per_step": 1e3 * dtl / late_steps,
                    "host_enqueue_ms_per_step": 1e3 * (th - tl) / late_steps,
                    "samples_per_sec": (c1["samples"] - c0["samples"]) / dtl,
                    "kept_samples_per_step": (c1["samples"] - c0["samples"]) / late_steps,
                    "marched_samples_per_step": (c1["marched"] - c0["marched"]) / late_steps,
                    "rays_per_step": (c1["rays"] - c0["rays"]) / late_steps}
            k += late_steps
            continue
        tr.train_step()
        k += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    c = tr.counters()
    final_loss = float(tr.last["loss"])
    finite = all(bool(torch.isfinite(p).all()) for p in model.parameters())
    test = SyntheticBlender(n_images=test_views, w=test_res, h=test_res, device=dev, seed=12345)  # unseen cameras
    model.eval()
    psnrs = []
    with torch.no_grad():
        for i in range(test_views):
            o, d = get_rays(test.directions.view(-1, 3), test.all_c2w[i:i + 1].expand(test_res * test_res, -1, -1))
            rays = torch.cat([o, torch.nn.functional.normalize(d, p=2, dim=-1)], -1)
            comp = render_rays(tr.fused, rays)["comp_rgb"]
            fg = test.all_fg_masks[i].view(-1, 1)
            gt = test.all_images[i].view(-1, 3) * fg + (1 - fg)
            psnrs.append(float(-10.0 * torch.log10(torch.mean((comp.to(dev).clamp(0, 1) - gt) ** 2))))
    whole = {"steps": n_steps, "seconds": dt, "ms_per_step": 1e3 * dt / n_steps, "samples_per_sec": c["samples"] / dt,
             "rays_per_sec": c["rays"] / dt, "kept_samples_per_step": c["samples"] / n_steps,
             "marched_samples_per_step": c["marched"] / n_steps, "truncated_launches": c["truncated"],
             "final_train_loss": final_loss, "parameters_finite": finite, "test_psnr": sum(psnrs) / len(psnrs),
             "test_psnr_per_view": psnrs, "test_views": f"{test_views} unseen {test_res}x{test_res} views of the procedural scene",
             "note": "fresh model, the reference's 20,000-step schedule, wall time of the whole loop (the two synchronisations of "
                     "the late-regime window included), rank 0, one GPU; procedural lego-like scene (no dataset on the box)"}
    del tr, model
    torch.cuda.empty_cache()
    return whole, late


def side_measurement(name, timeout=600):
    """run bench.<name>(cuda:0) in a child process and return its JSON: a side measurement must not be able to take the
    headline line down (a GPU fault kills the process it happens in)"""
    code = (f"import json, sys, torch; sys.argv = ['bench.py']; import bench; "
            f"print('SIDE_JSON ' + json.dumps(bench.{name}(torch.device('cuda', 0))))")
    try:
        p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=timeout,
                           env=dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", "")))
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("SIDE_JSON ")]
        if p.returncode != 0 or not line:
            return {"error": f"rc={p.returncode}: " + (p.stderr or p.stdout)[-300:]}
        return json.loads(line[-1][len("SIDE_JSON "):])
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)[:300]}


def exchange_report(tr, world, n_steps=32):
    """the gradient exchange of 32 further steps, timed with HIP events on the communication stream: the small fp32
    all-reduce, then per table range reduce-scatter (bf16) / AdamW on the shard / all-gather of the fp16 image; plus how
    long the step's own stream waits for the exchange after its last table-backward launch (the exposed part)"""
    sh = tr.sharded
    tr.comm_timings = {}
    exposed = []
    for _ in range(n_steps):
        tr.train_step()
        x = getattr(tr, "_xchg", None)
        if x is not None and getattr(sh, "_done_timed", None) is not None:
            torch.cuda.synchronize()
            exposed.append(x["events"][len(x["groups"]) - 1].elapsed_time(sh._done_timed))
    torch.cuda.synchronize()
    evs, small = tr.comm_timings.get("events", []), tr.comm_timings.get("small", [])
    tr.comm_timings = None
    rs = sum(e[0].elapsed_time(e[1]) for e in evs) / n_steps
    ad = sum(e[1].elapsed_time(e[2]) for e in evs) / n_steps
    ag = sum(e[2].elapsed_time(e[3]) for e in evs) / n_steps
    sm = sum(e[0].elapsed_time(e[1]) for e in small) / n_steps
    wire = sh.wire_bytes
    return {"small_all_reduce_ms": sm, "reduce_scatter_ms": rs, "sharded_adamw_ms": ad, "all_gather_ms": ag,
            "exposed_after_last_backward_launch_ms": (sum(exposed) / len(exposed)) if exposed else None,
            "ranges_per_step": len(evs) / n_steps, "table_ranges": {str(i): sh.ranges(m) for i, m in enumerate(sh.modules)
                                                                    if sh.ranges(m)},
            "level_groups": getattr(tr, "_xchg", None) and tr._xchg["groups"], "algo": sh.algo,
            "transport": str(sh.transport), "backend": dist.get_backend(), "wire_bytes_per_gpu_per_step": wire,
            "per_link_GBps": (wire / max(world - 1, 1)) / max((rs + ag) * 1e-3, 1e-9) / 1e9,
            "note": "algo ring = RCCL reduce_scatter_tensor / all_gather_into_tensor; a2a = pairwise over the xGMI mesh "
                    "(NSR_EXCHANGE_ALGO); per_link = bytes one of the P-1 links carries per step / (reduce_scatter + all_gather "
                    "time); the table backward writes bf16 straight into the send buffer, finest levels first, and the "
                    "exchange of a range starts behind its level group's event (nsr/parallel.py, csrc/step.hip)"}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_spawn(n_ranks):
    """`python bench.py --gpus N` from a bare interpreter: re-launch under torch.distributed.run, one rank per GPU
    (reference launch.py:93-107 lets Lightning spawn its DDP ranks the same way)"""
    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        raise SystemExit("bench.py needs an MI355X: no GPU visible")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(n_ranks, 1))))
    if n_dev < n_ranks:  # e.g. a 1-GPU box: RCCL refuses two ranks per device -- gloo rendezvous, every rank on device 0
        env.update(NSR_DIST_BACKEND="gloo", NSR_FORCE_DEVICE0="1")
        print(f"bench.py: {n_ranks} ranks requested, {n_dev} GPU(s) visible: the ranks share device 0 over gloo (a smoke run "
              "of the multi-rank path, not a scaling measurement)", file=sys.stderr)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--setup-steps", type=int, default=300,
                    help="untimed steps in front of the warm-up that bring the dynamic ray count / occupancy grid to the "
                         "operating point (8,192 rays per step)")
    ap.add_argument("--burn-in-steps", type=int, default=1500,
                    help="steps of a throwaway model in front of everything (box warm-up: libraries, clocks); 0 = off")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the C3 / C4 / C5 side measurements")
    ap.add_argument("--no-boundary-path", action="store_true", help="skip the step through nsr.models.FusedNeRFModel")
    ap.add_argument("--no-whole-run", action="store_true",
                    help="skip the 20,000-step run of a fresh model (whole_run / late_regime blocks, ~10 s)")
    ap.add_argument("--whole-run-steps", type=int, default=20000, help="length of that run (configs/nerf-blender.yaml:96)")
    ap.add_argument("--no-pipeline", action="store_true", help="diagnostic: march in order instead of on the side stream")
    ap.add_argument("--graphs", action="store_true", help="replay each step from a captured HIP graph (measured slower)")
    ap.add_argument("--sync-steps", action="store_true",
                    help="diagnostic: the step variant that reads its sample counts back to the host (two syncs/step)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_spawn(args.gpus)
    # NSR_DIST_BACKEND=gloo NSR_FORCE_DEVICE0=1 lets two ranks share ONE GPU: a smoke test of the multi-rank code path on
    # a 1-GPU box (the real runs use nccl = RCCL over xGMI, one rank per GPU)
    backend = os.environ.get("NSR_DIST_BACKEND", "nccl")
    shared_device = bool(os.environ.get("NSR_FORCE_DEVICE0"))
    if shared_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import nsr
    from nsr.scene import SyntheticBlender
    from nsr.trainer import Trainer
    from nsr_hip import ops

    cfg = nsr.configs.get("nerf-blender")
    data = SyntheticBlender(n_images=100, w=800, h=800, device=dev, seed=0)
    if args.burn_in_steps > 0:
        # a fresh box runs its first process slowly for the first second or so (libraries paging in, host and GPU clocks
        # ramping: the same command line measured 0.63 ms/step as the first process on a box and 0.50 as the second): steps
        # of a THROWAWAY model on the same data, before the measured model exists -- nothing of it is reused
        torch.manual_seed(1)
        tmp = Trainer(nsr.build(cfg).to(dev).train(), data, cfg, rank=0, world_size=1, seed=1, async_mode=not args.sync_steps)
        for _ in range(args.burn_in_steps):
            tmp.train_step()
        torch.cuda.synchronize()
        del tmp
        torch.cuda.empty_cache()
    torch.manual_seed(42)
    model = nsr.build(cfg).to(dev).train()
    tr = Trainer(model, data, cfg, rank=rank, world_size=world, seed=42, async_mode=not args.sync_steps)
    tr.pipeline_march = not args.no_pipeline
    tr.use_graphs = args.graphs

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Untimed set-up IN FRONT OF the warm-up: the dynamic ray count (systems/nerf.py:93-95) and the pruning of the occupancy
    # grid need ~300 steps to reach the operating point BASELINE.json quotes the metric on (8,192 rays/step); without it a
    # short `--warmup 5 --steps 20` run would time the start-up transient (1,147 rays/step through a dense grid).  The
    # transient itself (steps 5-25 of the fresh model) is timed on the way and reported as a labelled side block.
    transient = None
    for k in range(args.setup_steps):
        if tr.async_mode and k in (5, 25):
            sync()
            c = tr.counters()
            if k == 5:
                tc0, tt0 = c, time.perf_counter()
            else:
                dtt = time.perf_counter() - tt0
                transient = {"steps": [5, 25], "ms_per_step": 1e3 * dtt / 20,
                             "samples_per_sec": (c["samples"] - tc0["samples"]) / dtt,
                             "kept_samples_per_step": (c["samples"] - tc0["samples"]) / 20,
                             "rays_per_step": (c["rays"] - tc0["rays"]) / 20,
                             "note": "start-up transient of a fresh model (few rays, dense grid), rank 0's own clock; "
                                     "not the operating point"}
        tr.train_step()
    for _ in range(args.warmup):
        tr.train_step()
    sync()
    # asynchronous mode replays captured HIP graphs in the timed region (events cannot be recorded into a graph): the
    # per-kernel HIP-event timings are taken right after it, over further eager steps of the same run
    if not tr.async_mode:
        ops.profile_begin()
    c0 = tr.counters() if tr.async_mode else None  # device-side totals (reading them synchronises: outside the clock)
    sync()
    t0 = time.perf_counter()
    n_samples = n_rays = n_marched = 0
    for _ in range(args.steps):
        st = tr.train_step()
        if not tr.async_mode:
            n_samples += st["n_samples"]
            n_rays += st["n_rays"]
    t_enqueued = time.perf_counter() - t0  # the host has queued every step (it never waits inside one)
    sync()
    dt = time.perf_counter() - t0
    prof = ops.profile_end() if not tr.async_mode else {}
    prof_sep = None
    if tr.async_mode:
        c1 = tr.counters()
        n_samples, n_rays = c1["samples"] - c0["samples"], c1["rays"] - c0["rays"]
        n_marched = c1["marched"] - c0["marched"]
        if c1["truncated"] != c0["truncated"]:
            raise SystemExit("sample buffers overflowed inside the timed region: the measurement is invalid")
        # per-kernel durations: 64 further EAGER steps with the pooled HIP events of the C orchestration, recorded on
        # the stream each kernel is launched on; then 32 steps with the Python-side phase scopes
        n_prof = 8 if shared_device else 64  # (two ranks sharing a GPU over gloo stage every exchange through the host)
        ops.profile_begin(native_only=True)
        for _ in range(n_prof):
            tr.train_step()
        prof = ops.profile_end()
        c2 = tr.counters()
        live_m, live_s = c2["marched"] - c1["marched"], c2["samples"] - c1["samples"]
        live = {"hashgrid_forward": live_m, "mlp_forward_h1": live_m, "hashgrid_backward_params": live_s,
                "hashgrid_backward_bin": live_s, "hashgrid_backward_dense": live_s, "mlp_forward_h2": live_s, "mlp_backward_h1": live_s,
                "mlp_backward_h2": live_s}
        prof = {k: ((v[0], v[1], float(live[k])) if k in live else v) for k, v in prof.items()}
        ops.profile_begin()
        for _ in range(n_prof // 2):
            tr.train_step()
        for k, v2 in ops.profile_end().items():
            if k not in prof:
                prof[k] = v2
        # the table backward WITHOUT the optimizer folded into it (gradient store + separate AdamW kernel): 32 more steps,
        # so that the roofline line can also be read against the round-1 definition of the operation
        if world == 1 and tr.fuse_table_update:
            tr.fuse_table_update = False
            c3 = tr.counters()
            ops.profile_begin(native_only=True)
            for _ in range(32):
                tr.train_step()
            prof_sep = ops.profile_end()
            prof_sep["_samples"] = tr.counters()["samples"] - c3["samples"]
            tr.fuse_table_update = True

    comm = None
    if world > 1 and getattr(tr, "sharded", None) is not None:
        comm = exchange_report(tr, world, 4 if shared_device else 32)

    # steady state of the dynamic ray count (8192-ray cap reached, grid pruned): whatever --warmup / --steps were, run on
    # until >= 300 steps are behind and time 200 more -- OUTSIDE the driver-controlled region, reported beside `value`
    steady = None
    if tr.async_mode and not shared_device and not os.environ.get("NSR_BENCH_NO_STEADY"):
        while tr.global_step < 600:
            tr.train_step()
        sync()
        s0 = tr.counters()
        sync()
        ts = time.perf_counter()
        for _ in range(200):
            tr.train_step()
        sync()
        dts = time.perf_counter() - ts
        s1 = tr.counters()
        st = torch.tensor([dts, float(s1["samples"] - s0["samples"]), float(s1["rays"] - s0["rays"]),
                           float(s1["marched"] - s0["marched"])], dtype=torch.float64, device=dev)
        if world > 1:
            tm = st[:1].clone()
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            dist.all_reduce(st[1:], op=dist.ReduceOp.SUM)
            st[0] = tm[0]
        dts, ss, sr, sm = (float(v) for v in st.tolist())
        steady = {"steps_before": int(tr.global_step) - 200, "timed_steps": 200, "ms_per_step": 1e3 * dts / 200,
                  "samples_per_sec": ss / dts, "train_rays_per_sec": sr / dts, "kept_samples_per_step_per_gpu": ss / 200 / world,
                  "marched_samples_per_step_per_gpu": sm / 200 / world, "rays_per_step_per_gpu": sr / 200 / world,
                  "note": "same run, same trainer, after the driver-timed region: >= 600 steps behind, 200 steps timed with the "
                          "same barrier + synchronize bracket (max over ranks)"}

    tot = torch.tensor([dt, float(n_samples), float(n_rays)], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = tot[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot[1:], op=dist.ReduceOp.SUM)
        tot[0] = tmax[0]
    dt, n_samples, n_rays = (float(v) for v in tot.tolist())

    # Same-process A/B of the step's current forms against the round-4 forms of the same kernels (nsr.trainer.ROUND4_FORMS:
    # two data-gradient launches, one wave per ray compositing, scan + copy, events recorded behind kernels, the step's stream
    # waiting for the MLP optimizer in front of the encode): windows of 160 steps, interleaved A B B A, same trainer, same
    # regime -- box-to-box variation (+-5 % on this pool) cancels, which it does not between rounds
    forms_ab = None
    if world == 1 and tr.async_mode and not shared_device and not os.environ.get("NSR_BENCH_NO_FORMS_AB"):
        from nsr.trainer import ROUND4_FORMS, CURRENT_FORMS, set_step_forms
        acc = {"round4_forms": [], "current_forms": []}
        for name in ("round4_forms", "current_forms", "current_forms", "round4_forms"):
            set_step_forms(tr, ROUND4_FORMS if name == "round4_forms" else CURRENT_FORMS)
            for _ in range(16):
                tr.train_step()
            sync()
            a0, ta = tr.counters(), time.perf_counter()
            for _ in range(160):
                tr.train_step()
            sync()
            dta = time.perf_counter() - ta
            a1 = tr.counters()
            acc[name].append((1e3 * dta / 160, (a1["samples"] - a0["samples"]) / 160))
        set_step_forms(tr, CURRENT_FORMS)
        forms_ab = {k: {"ms_per_step": [round(x[0], 4) for x in v], "kept_samples_per_step": [round(x[1]) for x in v],
                        "mean_ms_per_step": sum(x[0] for x in v) / len(v)} for k, v in acc.items()}
        forms_ab["current_over_round4"] = forms_ab["current_forms"]["mean_ms_per_step"] / forms_ab["round4_forms"]["mean_ms_per_step"]
        forms_ab["at_step"] = int(tr.global_step)

    whole = late = None
    if (world == 1 and tr.async_mode and not shared_device and not args.no_whole_run
            and not os.environ.get("NSR_BENCH_NO_WHOLE_RUN")):
        whole, late = whole_run(dev, data, cfg, n_steps=args.whole_run_steps, late_at=min(10000, args.whole_run_steps // 2))

    async_mode, fuse_table_update = tr.async_mode, tr.fuse_table_update
    n_table_params = tr.fused.ewn.grid_desc.n_entries * tr.fused.ewn.grid_desc.n_features
    final_loss = float(tr.last["loss"])  # (a LazyLoss of the most recent step: read before anything else runs)
    others = None
    if not args.no_other_workloads:
        del tr, model  # (the NeuS trainers allocate their own tables)
        torch.cuda.empty_cache()
        others = other_workloads(dev, rank, world, sync, 4 if shared_device else 60)

    if rank == 0:
        # dominant kernel: the hash-grid launch class with the largest total time in the timed region
        roof = None
        kern = {}
        phase_steps = (4 if shared_device else 32) if async_mode else args.steps
        phases = {k[6:]: round(v[0] / phase_steps, 4) for k, v in prof.items() if k.startswith("phase:")}
        for name, (ms_total, launches, units) in prof.items():
            if name.startswith("phase:"):
                continue
            kern[name] = {"launches": launches, "avg_us": 1e3 * ms_total / max(launches, 1),
                          "units_per_launch": units / max(launches, 1)}
        # the table backward = item binning (on the main pass's helper stream, overlapped) + accumulation: one operation
        tb_pieces = None
        if "hashgrid_backward_bin" in prof and "hashgrid_backward_params" in prof:
            acc_ms, launches, units = prof["hashgrid_backward_params"]
            tb_pieces = {"binning_helper_stream_us": 1e3 * prof["hashgrid_backward_bin"][0] / max(launches, 1),
                         "accumulate_main_stream_us": 1e3 * acc_ms / max(launches, 1)}
            prof["hashgrid_backward_params"] = (acc_ms + prof["hashgrid_backward_bin"][0], launches, units)
        cand = {k: v for k, v in prof.items() if k.startswith("hashgrid") and k not in ("hashgrid_backward_bin", "hashgrid_backward_dense")}
        if cand:
            name = max(cand, key=lambda k: cand[k][0])
            ms_total, launches, units = cand[name]
            bps = ENC_FWD_BYTES_PER_SAMPLE if name == "hashgrid_forward" else ENC_BWD_BYTES_PER_SAMPLE
            # one GPU: AdamW on the table is applied inside the table backward (csrc/hashgrid.hip OwnerAdam) -- the launch
            # then also moves the optimizer's bytes: p, m, v read + p, m, v, fp16 image written per table parameter
            fused_opt = bool(name == "hashgrid_backward_params" and world == 1 and async_mode and fuse_table_update)
            opt_bytes = ADAM_TABLE_BYTES_PER_PARAM * n_table_params if fused_opt else 0
            if fused_opt:
                bps = ENC_BWD_FUSED_INPUT_BYTES_PER_SAMPLE
            per_launch = bps * units / launches + opt_bytes
            achieved = per_launch * launches / (ms_total * 1e-3) / 1e9
            traffic, traffic_note = pmc_traffic(name, units / launches, args.warmup, args.steps)
            frac = achieved / HBM_PEAK_GBS
            if traffic is not None and traffic < per_launch:
                frac, traffic_note = None, traffic_note + "; measured traffic below the algorithmic bytes: fraction withheld"
            roof = {"bound": "hbm", "kernel": name + ("+adamw(table)" if fused_opt else ""), "achieved": achieved,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frac, "traffic": traffic, "traffic_source": traffic_note,
                    "algorithmic_bytes_per_sample": bps, "samples_per_launch": units / launches,
                    "optimizer_bytes_per_launch": opt_bytes, "algorithmic_bytes_per_launch": per_launch,
                    "avg_launch_us": 1e3 * ms_total / launches,
                    "pieces": tb_pieces if name == "hashgrid_backward_params" else None}
            if fused_opt and prof_sep and "hashgrid_backward_params" in prof_sep:
                ms_b = prof_sep["hashgrid_backward_params"][0] + prof_sep.get("hashgrid_backward_bin", (0.0,))[0]
                n_l = prof_sep["hashgrid_backward_params"][1]
                ach_b = ENC_BWD_BYTES_PER_SAMPLE * prof_sep["_samples"] / (ms_b * 1e-3) / 1e9
                roof["separate_optimizer"] = {
                    "what": "32 further steps with the gradient stored and AdamW as its own kernel (the round-1 operation: "
                            "item binning + accumulation only)", "avg_launch_us": 1e3 * ms_b / max(n_l, 1),
                    "samples_per_launch": prof_sep["_samples"] / max(n_l, 1), "achieved": ach_b,
                    "frac": ach_b / HBM_PEAK_GBS}
        # what north_star asks to be evidenced beside the dominant kernel: HBM GB/s on the hash gather, MFMA utilisation on the
        # fused fp16 MLP.  Durations are THIS run's HIP-event averages; counters come from the PMC passes of the round
        # (tools/secondary_pmc.sh -> profiles/r06_secondary_pmc.json, 2^18 ray-coherent samples: labelled, never mixed into `value`)
        secondary = None
        try:
            sp_name = next((f for f in ("r06_secondary_pmc.json", "r04_secondary_pmc.json")
                            if os.path.exists(os.path.join(ROOT, "profiles", f))), None)
            sp = os.path.join(ROOT, "profiles", sp_name) if sp_name else ""
            pm = json.load(open(sp))["E2"]["kernels"] if sp_name else {}
            pick = lambda pre: next((v for k, v in pm.items() if k.startswith(pre)), {})  # noqa: E731
            gk, fk, dk, wk = pick("k_grid_forward"), pick("k_mlp_forward"), pick("k_mlp_dgrad"), pick("k_mlp_wgrad")
            secondary = {}
            if "hashgrid_forward" in kern:
                k = kern["hashgrid_forward"]
                ach = ENC_FWD_BYTES_PER_SAMPLE * k["units_per_launch"] / (k["avg_us"] * 1e-6) / 1e9
                secondary["hash_gather"] = {
                    "kernel": "k_grid_forward_pair", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes_per_sample": ENC_FWD_BYTES_PER_SAMPLE,
                    "samples_per_launch": k["units_per_launch"], "avg_launch_us": k["avg_us"],
                    "pmc_2e18_coherent": {c: gk.get(c) for c in ("fetch_MB", "write_MB", "l2_hit_rate")} or None,
                    "note": "588 B/sample counts every corner gather as if it came from HBM: the 24 MB table lives in the L2s / "
                            "the Infinity Cache (PMC: fabric fetch per launch far below it), so the kernel is bound by L2 "
                            "request rate, not by HBM"}
            for name, key, pmk, flops in (("mlp_forward_density", "mlp_forward_h1", fk, 2 * (32 * 64 + 64 * 16)),
                                          ("mlp_forward_color", "mlp_forward_h2", fk, 2 * (32 * 64 + 64 * 64 + 64 * 16))):
                if key in kern:
                    k = kern[key]
                    tf = flops * k["units_per_launch"] / (k["avg_us"] * 1e-6) / 1e12
                    secondary[name] = {"bound": "mfma", "achieved": tf, "peak": 2500.0, "unit": "TFLOP/s (fp16 dense)",
                                       "frac": tf / 2500.0, "avg_launch_us": k["avg_us"], "samples_per_launch": k["units_per_launch"],
                                       "pmc_mfma_busy_per_wave_cycle": pmk.get("mfma_busy_per_wave_cycle"),
                                       "note": "a 64-wide MLP moves 160-290 B per sample for 6-14 kFLOP: HBM-stream bound by two "
                                               "orders of magnitude below the MFMA peak; the MFMA pipe's busy share is the PMC figure"}
            secondary["mlp_backward_pmc_mfma_busy_per_wave_cycle"] = {"k_mlp_dgrad": dk.get("mfma_busy_per_wave_cycle"),
                                                                      "k_mlp_wgrad": wk.get("mfma_busy_per_wave_cycle")}
            secondary["pmc_source"] = f"profiles/{sp_name}" if pm else None
        except Exception as e:  # noqa: BLE001
            secondary = {"error": repr(e)[:200]}
        res = {
            "metric": "hash-encoded MLP samples/sec, full training step (march + encode + MLP + composite, fwd+bwd, "
                      "AdamW), nerf-blender lego config",
            "value": n_samples / dt, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": "nerf-blender (lego-like procedural scene): HashGrid L=16 T=2^19 F=2 + fused "
                                   "64-wide MLPs (1+2 hidden), dynamic <=8192 rays/step targeting 2^18 samples/step, "
                                   "100x800x800 views", "parallelism": f"ray-sharded dp{world}"},
            "train_rays_per_sec": n_rays / dt, "samples_per_step_per_gpu": n_samples / args.steps / world,
            "rays_per_step_per_gpu": n_rays / args.steps / world, "final_loss": final_loss,
            "regime": {"burn_in_steps_of_a_throwaway_model": args.burn_in_steps, "setup_steps": args.setup_steps,
                       "warmup_steps": args.warmup, "timed_steps": args.steps,
                       "kept_samples_per_step": n_samples / args.steps / world,
                       "marched_samples_per_step": (n_marched / args.steps) if n_marched else None,
                       "rays_per_step": n_rays / args.steps / world,
                       "note": "the operating point (8192-ray cap reached, grid pruned) needs ~300 steps: they run as untimed "
                               "set-up in front of the warm-up (--setup-steps); the start-up transient is the `transient` block"},
            "transient": transient,
            "host_enqueue_ms_per_step": 1e3 * t_enqueued / args.steps,
            "steady_state": steady, "step_forms_ab": forms_ab, "whole_run": whole, "late_regime": late,
            "roofline": roof, "roofline_secondary": secondary, "kernels": kern, "phase_ms_per_step": phases, "gradient_exchange": comm,
        }
        if shared_device:
            res["note"] = (f"{world} ranks SHARE one GPU over gloo (fewer GPUs than ranks on this box): a smoke run of the "
                           "multi-rank path, not a scaling measurement")
        if world == 1 and not args.no_boundary_path:
            res["boundary_path"] = side_measurement("boundary_path")
            res["boundary_path_neus"] = side_measurement("boundary_path_neus")
            res["modular_path"] = side_measurement("modular_path")
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
        res["other_workloads"] = others
        if os.environ.get("NSR_BENCH_REGIME_OUT"):  # the PMC passes record the regime they ran in (tools/pmc_traffic.py)
            reg = dict(res["regime"], roofline_units_per_launch={k: v["units_per_launch"] for k, v in kern.items()
                                                                 if k.startswith("hashgrid")},
                       # per-step kernels: the dispatch ordinals of the 64 steps the per-kernel durations are taken on
                       roofline_dispatch_window=[args.burn_in_steps + args.setup_steps + args.warmup + args.steps,
                                                 args.burn_in_steps + args.setup_steps + args.warmup + args.steps + 64])
            json.dump(reg, open(os.environ["NSR_BENCH_REGIME_OUT"], "w"))
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
