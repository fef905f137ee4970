"""Host-side profile (cProfile, own time) of NeuSTrainer.train_step at the reference's operating point: how much of the step the
host spends queueing it, and how long it waits for the marcher's sample count (nsr_hip.ops._spin_until_changed = GPU-bound time).
    python tools/neus_host_profile.py neus-blender|neus-dtu|neuralangelo [steps]"""
import cProfile, io, json, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd")]
import torch
import nsr
from nsr.fused_neus import NeuSTrainer
from nsr.scene import SyntheticBlender

name = sys.argv[1] if len(sys.argv) > 1 else "neus-blender"
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
lam = {"neus-blender": {"lambda_rgb_mse": 10.0, "lambda_rgb_l1": 0.0, "lambda_mask": 0.1, "lambda_eikonal": 0.1},
       "neus-dtu": {"lambda_rgb_mse": 0.0, "lambda_rgb_l1": 1.0, "lambda_mask": 0.0, "lambda_eikonal": 0.1},
       "neuralangelo": {"lambda_rgb_mse": 0.0, "lambda_rgb_l1": 1.0, "lambda_mask": 0.1, "lambda_eikonal": 0.1}}[name]
torch.manual_seed(7)
cfg = nsr.configs.get(name)
data = SyntheticBlender(n_images=20, w=400, h=400, device="cuda", seed=0, environment=bool(cfg["learned_background"]))
data.all_c2w[:, :, 3] *= float(cfg["radius"]) / 1.5
model = nsr.build(cfg).to("cuda").train()
tr = NeuSTrainer(model, data, cfg, lam, config_name=name)
if name == "neuralangelo":
    tr.global_step = 12000
for _ in range(n_steps):
    tr.train_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n_steps):
    tr.train_step()
torch.cuda.synchronize()
plain = 1e3 * (time.perf_counter() - t0) / n_steps
pr = cProfile.Profile()
pr.enable()
for _ in range(n_steps):
    tr.train_step()
pr.disable()
torch.cuda.synchronize()
buf = io.StringIO()
st = pstats.Stats(pr, stream=buf)
st.sort_stats("tottime").print_stats(40)
st.sort_stats("cumulative").print_stats(45)
print(json.dumps({"config": name, "ms_per_step_unprofiled": plain, "steps": n_steps}))
print(buf.getvalue())
