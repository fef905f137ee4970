"""Worker of tests/test_gpu_two_ranks.py: TWO ranks sharing ONE GPU (gloo rendezvous, both on cuda:0 -- RCCL refuses two
ranks per device, the 8-GPU runs are the driver's) drive the product trainers' multi-rank code path end to end:
parameter broadcast, per-rank ray batches, bf16 reduce-scatter -> sharded AdamW -> fp16 all-gather for the hash tables
(nsr.parallel.ShardedAdamW), all-reduce for the small fp32 heads.  Prints one JSON line from rank 0."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "instant-nsr-pl_amd")]


def digests(model):
    """per tensor: (sum, sum of squares) of what the kernels read next step (fp16 image of tcnn tensors, fp32 of the rest)"""
    import tinycudann as tcnn
    out, seen = {}, set()
    for name, mod in model.named_modules():
        if isinstance(mod, tcnn.Module) and mod.params.numel():
            h = mod.half_params(mod.params).double()
            out[name + ".fp16"] = (float(h.sum()), float((h * h).sum()))
            seen.add(id(mod.params))
    for name, p in model.named_parameters():
        if id(p) not in seen and p.numel():
            d = p.detach().double()
            out[name] = (float(d.sum()), float((d * d).sum()))
    return out


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    import nsr
    from nsr.fused_neus import NeuSTrainer
    from nsr.scene import SyntheticBlender
    from nsr.trainer import Trainer
    report = {}
    for name in ("nerf-blender", "nerf-blender-async", "neus-dtu"):
        cfg = nsr.configs.get(name.replace("-async", ""))
        torch.manual_seed(100 + rank)  # different initial weights on purpose: the broadcast has to make them equal
        model = nsr.build(cfg).to(dev).train()
        data = SyntheticBlender(n_images=8, w=64, h=64, device=dev, seed=0, environment=bool(cfg.get("learned_background")))
        data.all_c2w[:, :, 3] *= float(cfg["radius"]) / 1.5
        if name.startswith("nerf-blender"):
            # "-async": the measured path of bench.py -- no host sync, the table backward writes bf16 into the exchange's send
            # buffer in two level groups, the MLP gradients go into the flattened small message (nsr_nerf_main_pass_exchange)
            tr = Trainer(model, data, cfg, rank=rank, world_size=world, seed=42, async_mode=name.endswith("-async"))
        else:
            tr = NeuSTrainer(model, data, cfg, {"lambda_rgb_l1": 1.0, "lambda_eikonal": 0.1}, config_name=name, rank=rank,
                             world_size=world, seed=42)
        assert tr.sharded is not None, "multi-rank trainers exchange the tables through ShardedAdamW"
        first = digests(model)
        counts = []
        for _ in range(6 if not name.endswith("-async") else 20):  # (the asynchronous run crosses a grid refresh at step 16)
            counts.append(int(tr.train_step()["n_samples"]))
        torch.cuda.synchronize()
        # the occupancy grids are synchronised behind every refresh (DDP broadcast_buffers semantics): identical on all ranks
        gsum = torch.stack([model.occupancy_grid._binary.sum().double(), model.occupancy_grid.occs.double().sum()])
        gall = [torch.empty_like(gsum) for _ in range(world)]
        dist.all_gather(gall, gsum)
        assert all(torch.equal(g, gall[0]) for g in gall), "occupancy grids drifted apart between the ranks"
        mine = digests(model)
        groups = None
        if name.endswith("-async"):
            assert tr._xchg is not None and len(tr._xchg["groups"]) == 2, "the asynchronous step exchanges in two level groups"
            groups = tr._xchg["groups"]
            # a checkpoint of a multi-rank run: the fp32 table lives in the owners' shards until state_dict() gathers it
            sd = tr.state_dict()
            full = sd["geometry.encoding_with_network.params"].float()
            img = model.geometry.encoding_with_network.half_params(model.geometry.encoding_with_network.params)
            assert torch.equal(full.half(), img), "gathered fp32 master != the fp16 image the kernels read"
        both = [None] * world
        dist.all_gather_object(both, {"digest": mine, "counts": counts})
        if rank == 0:
            a, b = both[0]["digest"], both[1]["digest"]
            assert a.keys() == b.keys()
            worst = max(abs(a[k][0] - b[k][0]) + abs(a[k][1] - b[k][1]) for k in a)
            moved = sum(1 for k in mine if mine[k] != first[k])
            report[name] = {"tensors": len(a), "replica_mismatch": worst, "tensors_moved": moved, "groups": groups,
                            "samples_rank0": both[0]["counts"], "samples_rank1": both[1]["counts"],
                            "finite": all(v[1] == v[1] and abs(v[1]) < 1e30 for v in a.values())}
        del tr, model, data
        torch.cuda.empty_cache()
    if rank == 0:
        print("TWO_RANK_REPORT " + json.dumps(report), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
