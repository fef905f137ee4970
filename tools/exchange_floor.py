"""What the multi-GPU exchange costs a step BEFORE any byte travels: the asynchronous NeRF trainer (and the NeuS trainers)
on a ONE-rank nccl process group with the sharded path forced (NSR_FORCE_SHARDED=1: bf16 send buffer written by the table
backward in two level groups, 4 one-rank RCCL collectives + the small all-reduce on the communication stream, AdamW as
its own sweep over the whole table) against the plain one-GPU step.  One JSON line.

    python tools/exchange_floor.py [nerf-blender|neus-blender|neus-dtu|neuralangelo ...]
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd")]
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
import nsr
from nsr.fused_neus import NeuSTrainer
from nsr.scene import SyntheticBlender
from nsr.trainer import Trainer

names = sys.argv[1:] or ["nerf-blender"]
res = {}
for name in names:
    cfg = nsr.configs.get(name)
    for sharded in (False, True):
        if sharded:
            os.environ["NSR_FORCE_SHARDED"] = "1"
        else:
            os.environ.pop("NSR_FORCE_SHARDED", None)
        torch.manual_seed(7)
        model = nsr.build(cfg).cuda().train()
        if name == "nerf-blender":
            data = SyntheticBlender(n_images=100, w=800, h=800, device="cuda", seed=0)
            tr = Trainer(model, data, cfg, async_mode=True)
            warm, steps = 400, 200
        else:
            data = SyntheticBlender(n_images=20, w=400, h=400, device="cuda", seed=0, environment=bool(cfg["learned_background"]))
            data.all_c2w[:, :, 3] *= float(cfg["radius"]) / 1.5
            tr = NeuSTrainer(model, data, cfg, {"lambda_rgb_l1": 1.0, "lambda_eikonal": 0.1}, config_name=name)
            if name == "neuralangelo":
                tr.global_step = 12000
            warm, steps = 100, 100
        assert (tr.sharded is not None) == sharded
        for _ in range(warm):
            tr.train_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.train_step()
        torch.cuda.synchronize()
        res[f"{name}:{'sharded_1rank_nccl' if sharded else 'one_gpu'}"] = round(1e3 * (time.perf_counter() - t0) / steps, 4)
        del tr, model, data
        torch.cuda.empty_cache()
dist.destroy_process_group()
print(json.dumps(res))
