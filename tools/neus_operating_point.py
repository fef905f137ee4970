"""One fused NeuS workload at the reference's operating point (dynamic ray count -> 2^18 samples / step) through
NeuSTrainer: ms / step, and how much of it the HOST spends queueing the step (time inside train_step, which holds one
device->host read of the sample count) -- tells a GPU-bound step from a host-bound one.  One JSON line.

    python tools/neus_operating_point.py neus-blender|neus-dtu|neuralangelo [steps [owner_large_from]]
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd")]
import torch
import nsr
from nsr.fused_neus import NeuSTrainer
from nsr.scene import SyntheticBlender

name = sys.argv[1] if len(sys.argv) > 1 else "neus-blender"
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
if len(sys.argv) > 3:  # developer switch: point count from which the table backward takes its 2^13 x 1024 configuration
    from nsr_hip import lib
    lib.nsr_hashgrid_owner_large_from(int(sys.argv[3]))
lam = {"neus-blender": {"lambda_rgb_mse": 10.0, "lambda_rgb_l1": 0.0, "lambda_mask": 0.1, "lambda_eikonal": 0.1},
       "neus-dtu": {"lambda_rgb_mse": 0.0, "lambda_rgb_l1": 1.0, "lambda_mask": 0.0, "lambda_eikonal": 0.1},
       "neuralangelo": {"lambda_rgb_mse": 0.0, "lambda_rgb_l1": 1.0, "lambda_mask": 0.1, "lambda_eikonal": 0.1}}[name]
dev = "cuda"
torch.manual_seed(7)
cfg = nsr.configs.get(name)
data = SyntheticBlender(n_images=20, w=400, h=400, device=dev, seed=0, environment=bool(cfg["learned_background"]))
data.all_c2w[:, :, 3] *= float(cfg["radius"]) / 1.5
model = nsr.build(cfg).to(dev).train()
tr = NeuSTrainer(model, data, cfg, lam, config_name=name)
if name == "neuralangelo":
    tr.global_step = 12000
for _ in range(n_steps):
    tr.train_step()
torch.cuda.synchronize()
import nsr_hip.ops as _ops
_ops.SPIN_SECONDS[0] = 0.0
t0, n, host, n_fg, n_bg, m_bg = time.perf_counter(), 0, 0.0, 0, 0, 0
for _ in range(n_steps):
    h0 = time.perf_counter()
    last = tr.train_step()
    host += time.perf_counter() - h0
    n += last["n_samples"] + last["n_samples_bg"]
    n_fg += last["n_samples"]; n_bg += last["n_samples_bg"]; m_bg += last.get("n_marched_bg", 0)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"config": name, "ms_per_step": 1e3 * dt / n_steps, "host_ms_in_train_step": 1e3 * host / n_steps, "host_ms_waiting_for_counts": 1e3 * _ops.SPIN_SECONDS[0] / n_steps,
                  "samples_per_step": n / n_steps, "fg_samples_per_step": n_fg / n_steps, "bg_kept_per_step": n_bg / n_steps,
                  "bg_marched_per_step": m_bg / n_steps, "rays_per_step": tr.train_num_rays, "samples_per_sec": n / dt}))
