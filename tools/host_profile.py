"""where does the HOST spend its time in an asynchronous training step? (cProfile over 300 steps)"""
import os, sys, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "instant-nsr-pl_amd"))
import torch, nsr
from nsr.scene import SyntheticBlender
from nsr.trainer import Trainer
torch.manual_seed(42)
cfg = nsr.configs.get("nerf-blender")
model = nsr.build(cfg).cuda().train()
small = bool(os.environ.get("NSR_HOST_PROFILE_SMALL"))  # a 12-image 400x400 set builds in seconds
data = SyntheticBlender(n_images=12 if small else 100, w=400 if small else 800, h=400 if small else 800, device="cuda", seed=0)
tr = Trainer(model, data, cfg, seed=42, async_mode=True)
for _ in range(300): tr.train_step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(300): tr.train_step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumtime").print_stats(30)
