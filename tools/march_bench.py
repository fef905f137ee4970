"""The brick marcher on a trained occupancy grid: one ray set of 8,192 rays (what the model-interface path marches per step)
and a window of 15 sets (what the asynchronous trainer marches per launch), by rays carried per wave; outputs compared bit for
bit against the 64-rays-per-wave launch.   python tools/march_bench.py [train_steps]"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd"), os.path.join(ROOT, "tools")]
import torch
import nsr
from nsr.scene import SyntheticBlender
from nsr.trainer import Trainer
from nsr.fused import prepare_train_rays
from nsr_hip import check, lib, ptr, stream_ptr, ops
from nerfacc import ContractionType
from kernel_microbench import median_us

if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    dev = torch.device("cuda", 0)
    torch.manual_seed(42)
    cfg = nsr.configs.get("nerf-blender")
    model = nsr.build(cfg).to(dev).train()
    data = SyntheticBlender(n_images=100, w=800, h=800, device=dev, seed=0)
    tr = Trainer(model, data, cfg, seed=42, async_mode=True)
    for _ in range(steps):
        tr.train_step()
    torch.cuda.synchronize()
    grid = model.occupancy_grid
    bricks = ops.grid_bricks(grid.binary)
    rx, ry, rz = (int(v) for v in grid.binary.shape)
    cap = int(lib.nsr_ray_march_capacity((ctypes.c_float * 6)(*[float(v) for v in grid._roi_host]), float(model.render_step_size)))
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    res = {"occupied_fraction": float(grid.binary.float().mean()), "cases": {}}
    for n_sets in (1, 4, 15):
        n = 8192 * n_sets
        rays, ro, rd, rgb, fg, bg, t_min, t_max = prepare_train_rays(data, n, gen, model, cfg["background_color"])
        counts = torch.empty(n, dtype=torch.int32, device=dev)
        scratch = torch.empty(n * cap * 2, dtype=torch.float32, device=dev)
        ref = None
        for rpw in (64, 16, "wave"):
            lib.nsr_ray_march_wave_mode(2 if rpw == "wave" else 0)
            lib.nsr_ray_march_rays_per_wave(64 if rpw == "wave" else rpw)
            def f():
                check(lib.nsr_ray_march_bricks_count(ptr(ro), ptr(rd), ptr(t_min), ptr(t_max), ptr(grid.roi_aabb), ptr(bricks), rx, ry, rz,
                                                     ContractionType.AABB.value, float(model.render_step_size), 0.0, ptr(counts),
                                                     ptr(scratch), cap, n, stream_ptr()), "march")
            us = median_us(f, 3, 10)
            f(); torch.cuda.synchronize()
            rows = scratch.view(n, cap, 2)
            valid = (torch.arange(cap, device=dev)[None, :] < counts[:, None].clamp(max=cap))[..., None].expand(-1, -1, 2)
            packed = torch.where(valid, rows, torch.zeros((), device=dev))
            if ref is None:
                ref = (counts.clone(), packed.clone())
            same = bool(torch.equal(counts, ref[0]))
            same_t = bool(torch.equal(packed, ref[1]))
            res["cases"][f"{n_sets}x8192:{rpw}"] = {"us": round(us, 1), "same_counts": same, "same_t0_t1_bits": same_t,
                                                    "marched": int(counts.sum())}
            del rows, valid, packed
        lib.nsr_ray_march_rays_per_wave(0)
        lib.nsr_ray_march_wave_mode(1)
    print(json.dumps(res))
