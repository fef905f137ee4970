"""the 256^3 background-grid refresh of neus-dtu (C4) in isolation: builds the model, trains 40 steps (so that the grid is
pruned and the density head is not at its initial state), then refreshes 6 times behind the warm-up (random quarter +
occupied cells, nerfacc 0.3.3 `_update`).  Prints ms per refresh; run under rocprofv3 --kernel-trace --stats for the
per-kernel split (tools/bg_refresh_profile.sh)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd")]
import torch
import nsr
from nsr.fused_neus import NeuSTrainer
from nsr.scene import SyntheticBlender

torch.manual_seed(7)
cfg = nsr.configs.get("neus-dtu")
data = SyntheticBlender(n_images=20, w=400, h=400, device="cuda", seed=0, environment=True)
data.all_c2w[:, :, 3] *= float(cfg["radius"]) / 1.5
model = nsr.build(cfg).cuda().train()
tr = NeuSTrainer(model, data, cfg, {"lambda_rgb_l1": 1.0, "lambda_eikonal": 0.1}, config_name="neus-dtu")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for _ in range(steps):
    tr.train_step()
torch.cuda.synchronize()
g = model.occupancy_grid_bg
occ = int(g.binary.sum())
t0 = time.perf_counter()
device = not os.environ.get("NSR_NEUS_TORCH_REFRESH")
for k in range(7):
    if k == 1:  # (the first call allocates the scratch buffers)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    if device:
        tr.fused.refresh_bg_occupancy_async(4096 + 16 * k, occ_thre=cfg.get("grid_prune_occ_thre_bg", 0.01))
    else:
        g.every_n_step(step=4096 + 16 * k, occ_eval_fn=tr.fused.bg_occ_eval_fn, occ_thre=cfg.get("grid_prune_occ_thre_bg", 0.01))
torch.cuda.synchronize()
print(json.dumps({"path": "device" if device else "torch", "ms_per_refresh": 1e3 * (time.perf_counter() - t0) / 6, "occupied_cells_before": occ,
                  "occupied_cells_after": int(g.binary.sum()), "cells": 256 ** 3}))
