"""Worker of tests/test_gpu_nccl_single_rank.py: ONE rank in an nccl (= RCCL) process group drives the trainers' multi-GPU
exchange (NSR_FORCE_SHARDED=1): bf16 reduce_scatter_tensor of the send buffer the table backward wrote, sharded AdamW, fp16
all_gather_into_tensor straight into the shadow tensor, the flattened fp32 all_reduce, all on the communication stream behind
HIP events recorded by the C step -- every RCCL call of the path with its real dtypes, views and streams (RCCL refuses two
ranks per device, so on a one-GPU box the collectives are one-rank ones).  Prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd")]


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    os.environ["NSR_FORCE_SHARDED"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    dev = torch.device("cuda", 0)
    import nsr
    from nsr.fused_neus import NeuSTrainer
    from nsr.scene import SyntheticBlender
    from nsr.trainer import Trainer
    report = {"backend": dist.get_backend()}
    for name in ("nerf-blender-async", "nerf-blender", "neus-dtu", "neuralangelo"):
        cfg = nsr.configs.get(name.replace("-async", ""))
        runs = {}
        # the sharded path with bf16 (the default wire format) and with fp32 gradients, then the plain one-GPU optimizer path
        for mode in ("bf16", "fp32", None):
            sharded = mode is not None
            if sharded:
                os.environ["NSR_FORCE_SHARDED"] = "1"
                os.environ["NSR_TRANSPORT"] = mode
            else:
                os.environ.pop("NSR_FORCE_SHARDED", None)
                os.environ.pop("NSR_TRANSPORT", None)
            torch.manual_seed(5)
            model = nsr.build(cfg).to(dev).train()
            data = SyntheticBlender(n_images=8, w=64, h=64, device=dev, seed=0, environment=bool(cfg.get("learned_background")))
            data.all_c2w[:, :, 3] *= float(cfg["radius"]) / 1.5
            if name.startswith("nerf-blender"):
                tr = Trainer(model, data, cfg, rank=0, world_size=1, seed=42, async_mode=name.endswith("-async"))
            else:
                tr = NeuSTrainer(model, data, cfg, {"lambda_rgb_l1": 1.0, "lambda_eikonal": 0.1}, config_name=name, seed=42)
            assert (tr.sharded is not None) == sharded
            if sharded:
                tr.comm_timings = {}
            for _ in range(5):
                last = tr.train_step()
            torch.cuda.synchronize()
            import tinycudann as tcnn
            tabs = [m for m in model.modules() if isinstance(m, tcnn.Module) and m.params.numel() > 100000]
            runs[mode] = [m.half_params(m.params).float().clone() for m in tabs]
            if mode == "bf16":
                sd = tr.state_dict()  # gathers the masters (a collective)
                ev = tr.comm_timings.get("events", [])
                rs = sum(e[0].elapsed_time(e[1]) for e in ev) / max(len(ev), 1)
                report[name] = {"ranges_timed": len(ev), "reduce_scatter_ms": rs, "keys": len(sd),
                                "groups": (len(tr._xchg["groups"]) if getattr(tr, "_xchg", None) else None)}
            del tr, model
        # same seeds, same batches.  fp32 on the wire: the sharded optimizer is the one-GPU optimizer up to the order of its fp32
        # sums; bf16: Adam's normalised step (eps 1e-15) turns the rounding of a near-zero gradient into a full +-lr step
        rel = lambda x, y: max(float((a - b).norm() / max(float(a.norm()), 1e-12)) for a, b in zip(x, y))  # noqa: E731
        report[name].update(max_abs_diff=max(float((a - b).abs().max()) for a, b in zip(runs["bf16"], runs[None])),
                            rel_l2_diff=rel(runs["bf16"], runs[None]), rel_l2_diff_fp32_transport=rel(runs["fp32"], runs[None]),
                            finite=all(bool(torch.isfinite(a).all()) for a in runs["bf16"] + runs["fp32"]))
    dist.destroy_process_group()
    print(json.dumps(report))


if __name__ == "__main__":
    main()
