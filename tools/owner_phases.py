"""Where an owner workgroup of the table backward spends its time, per level: needs the debug build of the library
(-DNSR_OWN_TIMING: s_memrealtime stamps at the phase boundaries, summed over workgroups into a device array), e.g.
    NSR_EXTRA_FLAGS=-DNSR_OWN_TIMING bash instant-nsr-pl_amd/csrc/build.sh build/variants/timing
    NSR_HIP_LIB=build/variants/timing/libnsr_hip.so python tools/owner_phases.py neuralangelo|neus-blender|neus-dtu|nerf
Prints per level: workgroups, items, and the mean microseconds per workgroup of: set-up + LDS clear, directory scan,
item loop, write-out / AdamW."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd")]
import torch
import nsr
import nsr_hip
from nsr.scene import SyntheticBlender

name = sys.argv[1] if len(sys.argv) > 1 else "neuralangelo"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
h = ctypes.CDLL(nsr_hip.LIB_PATH)
buf = (ctypes.c_ulonglong * (32 * 8))()
dev = "cuda"
torch.manual_seed(7)
if name == "nerf":
    from nsr.trainer import Trainer
    cfg = nsr.configs.get("nerf-blender")
    data = SyntheticBlender(n_images=100, w=800, h=800, device=dev, seed=0)
    tr = Trainer(nsr.build(cfg).to(dev).train(), data, cfg, seed=42, async_mode=True)
    warm = 1500
else:
    from nsr.fused_neus import NeuSTrainer
    lam = {"neus-blender": {"lambda_rgb_mse": 10.0, "lambda_rgb_l1": 0.0, "lambda_mask": 0.1, "lambda_eikonal": 0.1},
           "neus-dtu": {"lambda_rgb_mse": 0.0, "lambda_rgb_l1": 1.0, "lambda_mask": 0.0, "lambda_eikonal": 0.1},
           "neuralangelo": {"lambda_rgb_mse": 0.0, "lambda_rgb_l1": 1.0, "lambda_mask": 0.1, "lambda_eikonal": 0.1}}[name]
    cfg = nsr.configs.get(name)
    data = SyntheticBlender(n_images=20, w=400, h=400, device=dev, seed=0, environment=bool(cfg["learned_background"]))
    data.all_c2w[:, :, 3] *= float(cfg["radius"]) / 1.5
    tr = NeuSTrainer(nsr.build(cfg).to(dev).train(), data, cfg, lam, config_name=name)
    if name == "neuralangelo":
        tr.global_step = 12000
    warm = 100
for _ in range(warm):
    tr.train_step()
torch.cuda.synchronize()
for large in (0, 1):
    h.nsr_debug_owner_timing(buf, large)  # clear
for _ in range(steps):
    tr.train_step()
torch.cuda.synchronize()
out = {"config": name, "steps": steps}
for large, tag in ((0, "small_2^11x256"), (1, "large_2^13x1024")):
    h.nsr_debug_owner_timing(buf, large)
    rows = []
    for lv in range(16):
        u, t_setup, t_scan, t_items, t_out, n_items = (buf[lv * 8 + k] for k in range(6))
        if u == 0:
            continue
        us = lambda t: round(t / u / 100.0, 2)  # 100 MHz ticks -> us per workgroup  # noqa: E731
        rows.append({"level": lv, "wgs_per_step": round(u / steps, 1), "items_per_wg": round(n_items / u, 1),
                     "setup_clear_us": us(t_setup), "dir_scan_us": us(t_scan), "item_loop_us": us(t_items), "write_out_us": us(t_out),
                     "wg_total_us": us(t_setup + t_scan + t_items + t_out),
                     "ns_per_item": round(10.0 * t_items / max(n_items, 1), 2)})
    if rows:
        out[tag] = rows
print(json.dumps(out))
