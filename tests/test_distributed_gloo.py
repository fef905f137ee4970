"""Multi-process logic of the ray-sharded data-parallel path on CPU: world_size 2, gloo, 127.0.0.1.
(The same functions run over RCCL on the MI355X node; the GPU kernels themselves are covered by the -m gpu tests.)"""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out, half_transport=False):
    for p in (ROOT, os.path.join(ROOT, "instant-nsr-pl_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nsr.parallel import all_reduce_gradients, broadcast_parameters, rank_world, shard_seed
    assert rank_world() == (rank, world, rank)
    torch.manual_seed(100 + rank)  # replicas start DIFFERENT on purpose: broadcast must fix that
    model = torch.nn.ModuleDict({"table": torch.nn.Embedding(70000, 2), "mlp": torch.nn.Linear(8, 3)})
    broadcast_parameters(model)
    w0 = model["table"].weight.detach().clone()
    # rank-sharded "rays": every rank draws its own batch
    g = torch.Generator().manual_seed(shard_seed(42, rank))
    idx = torch.randint(0, 70000, (256,), generator=g)
    x = torch.randn(256, 8, generator=g)
    loss = model["table"](idx).pow(2).mean() + model["mlp"](x).pow(2).mean()
    loss.backward()
    local = [p.grad.clone() for p in model.parameters()]
    n_bytes = all_reduce_gradients(list(model.parameters()), half_transport=half_transport)
    torch.save({"w0": w0, "idx": idx, "x": x, "local": local, "avg": [p.grad.clone() for p in model.parameters()],
                "n_bytes": n_bytes}, os.path.join(out, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_ray_sharded_gradient_all_reduce(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"rank{k}.pt") for k in range(world)]
    assert torch.equal(r[0]["w0"], r[1]["w0"])                       # replicas identical after broadcast
    assert not torch.equal(r[0]["idx"], r[1]["idx"])                 # ... but their ray batches differ
    for k in range(len(r[0]["avg"])):
        mean = (r[0]["local"][k] + r[1]["local"][k]) / 2
        assert torch.allclose(r[0]["avg"][k], mean, atol=1e-7) and torch.equal(r[0]["avg"][k], r[1]["avg"][k])
    assert r[0]["n_bytes"] == (70000 * 2 + 8 * 3 + 3) * 4            # table gradient + flattened small gradients
    # the averaged gradient equals the single-process gradient of the union batch (mean loss, equal shards)
    torch.manual_seed(0)
    model = torch.nn.ModuleDict({"table": torch.nn.Embedding(70000, 2), "mlp": torch.nn.Linear(8, 3)})
    # same initial weights as the broadcast replicas: only the table is checked against w0 here
    with torch.no_grad():
        model["table"].weight.copy_(r[0]["w0"])
    idx = torch.cat([r[0]["idx"], r[1]["idx"]])
    model["table"](idx).pow(2).mean().backward()
    assert torch.allclose(model["table"].weight.grad, r[0]["avg"][0], atol=1e-7)


def test_gradient_all_reduce_with_half_precision_transport(tmp_path):
    """the large (table) gradient travels as fp16 x 1024: same mean within fp16 rounding, half the bytes on the wire"""
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), True), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"rank{k}.pt") for k in range(world)]
    mean = (r[0]["local"][0] + r[1]["local"][0]) / 2
    assert torch.equal(r[0]["avg"][0], r[1]["avg"][0])                       # replicas stay identical
    assert torch.allclose(r[0]["avg"][0], mean, rtol=2e-3, atol=1e-7)        # 11-bit mantissa on the wire
    assert bool((r[0]["avg"][0] != 0).sum() == (mean != 0).sum())           # nothing underflowed
    for k in (1, 2):                                                         # small gradients stay fp32
        assert torch.allclose(r[0]["avg"][k], (r[0]["local"][k] + r[1]["local"][k]) / 2, atol=1e-7)
    assert r[0]["n_bytes"] == 70000 * 2 * 2 + (8 * 3 + 3) * 4


def test_shard_seed_and_single_process_noop():
    for p in (ROOT, os.path.join(ROOT, "instant-nsr-pl_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from nsr.parallel import all_reduce_gradients, shard_seed
    assert len({shard_seed(42, r) for r in range(8)}) == 8 and shard_seed(42, 0) == 42
    lin = torch.nn.Linear(2, 2)
    lin(torch.ones(1, 2)).sum().backward()
    assert all_reduce_gradients(list(lin.parameters())) == 0  # not initialised: no collective


def test_asynchronous_capacity_policy():
    """nsr.trainer.next_capacity: the host-side rule that sizes the sample buffers from lagged device statistics"""
    for p in (ROOT, os.path.join(ROOT, "instant-nsr-pl_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    try:
        from nsr.trainer import next_capacity
    except ImportError as e:  # the product package refuses to import without its HIP library
        import pytest
        pytest.skip(f"nsr not importable here: {e}")
    assert next_capacity(1 << 20, 0, 8192, 8192, False) == 1 << 20                 # no statistics yet: keep
    assert next_capacity(1 << 20, 300_000, 8192, 8192, False) == 458752           # > 2x too large: shrink to 1.5x, 16k granule
    assert next_capacity(458752, 300_000, 8192, 8192, False) == 458752            # comfortable: keep
    assert next_capacity(458752, 400_000, 8192, 8192, False) == 606208            # > 85 % full: grow
    assert next_capacity(458752, 900_000, 8192, 8192, True) == 1359872            # samples were dropped: grow to fit
    assert next_capacity(1 << 20, 300_000, 8192, 8192, True) == 1 << 20           # never shrink right after a drop
    assert next_capacity(2 << 20, 100_000, 512, 2048, False) == 606208            # ray count at 1/4 of its maximum: 4x room
    assert next_capacity(1 << 20, 100_000, 512, 2048, False) == 1 << 20           # ... which keeps a 1 Mi buffer
    assert next_capacity(1 << 20, 1000, 8192, 8192, False) == 65536               # floor


def test_lazy_loss_refuses_stale_reads():
    """nsr.trainer.LazyLoss: an asynchronous step's loss lives in that step's accumulator and must be read before the next"""
    import pytest
    for p in (ROOT, os.path.join(ROOT, "instant-nsr-pl_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    try:
        from nsr.trainer import LazyLoss
    except ImportError as e:
        pytest.skip(f"nsr not importable here: {e}")

    class FakeTrainer:
        global_step = 7

    tr = FakeTrainer()
    loss = LazyLoss(torch.tensor([6.0, 4.0]), tr)            # sum of smooth-L1 terms, number of valid rays
    assert abs(float(loss) - 0.5) < 1e-7 and abs(loss.item() - 0.5) < 1e-7 and bool(loss.isfinite())
    assert abs(float(LazyLoss(torch.tensor([0.0, 0.0]), tr))) == 0.0   # no valid ray: 0 / max(0, 1)
    tr.global_step = 8
    with pytest.raises(RuntimeError):
        float(loss)
