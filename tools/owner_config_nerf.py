"""The table backward's configuration / placement at the NeRF operating point: ms per step in two windows of one run.
    python tools/owner_config_nerf.py <large_from: points from which the 2^13 x 1024 configuration is taken> <placement 0..4>
(round 6: 400000 4 = the default 0.408-0.413 / 0.375 ms; 0 3 = large + claimed 0.440-0.443 / 0.395-0.399; 0 2 = large + striped 0.436-0.438 / 0.387-0.402)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd")]
import torch, nsr
from nsr_hip import lib
from nsr.scene import SyntheticBlender
from nsr.trainer import Trainer
thr = int(sys.argv[1]); place = float(sys.argv[2])
lib.nsr_hashgrid_owner_large_from(thr); lib.nsr_hashgrid_owner_tune(0, place)
cfg = nsr.configs.get("nerf-blender")
data = SyntheticBlender(n_images=100, w=800, h=800, device="cuda", seed=0)
tr = Trainer(nsr.build(cfg).to("cuda").train(), data, cfg, seed=42, async_mode=True)
for _ in range(700): tr.train_step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(400): tr.train_step()
torch.cuda.synchronize(); a = (time.perf_counter() - t0) / 400 * 1e3
for _ in range(3000): tr.train_step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(400): tr.train_step()
torch.cuda.synchronize(); b = (time.perf_counter() - t0) / 400 * 1e3
print(f"large_from={thr} placement={place}: steady(700-1100) {a:.4f} ms  late(4100-4500) {b:.4f} ms")
