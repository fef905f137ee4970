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


class _FakeTcnnModule(torch.nn.Module):
    """what ShardedAdamW needs from a tinycudann module: one flat fp32 ``params`` and ``adopt_shadow``"""

    def __init__(self, n, seed, n_network_params=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.params = torch.nn.Parameter(torch.randn(n, generator=g) * 0.1)
        self.params.grad = torch.zeros(n)
        self.n_network_params = n_network_params  # MLP weights in front of the table (NetworkWithInputEncoding)
        self.shadow = None

    def adopt_shadow(self, shadow):
        self.shadow = shadow

    def invalidate(self):
        self.shadow = None


def _sharded_worker(rank, world, port, out, transport):
    for p in (ROOT, os.path.join(ROOT, "instant-nsr-pl_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nsr.parallel import ShardedAdamW
    # [3072 MLP weights | ragged 70003-element table] (the last shard is short) + a small module (replicated, not sharded)
    mods = [_FakeTcnnModule(3072 + 70003, 7, n_network_params=3072), _FakeTcnnModule(7168, 8)]
    # the table is exchanged in three ranges (cut points rounded up to world x 8 elements), highest offsets first
    opt = ShardedAdamW(mods, lr=0.01, transport=getattr(torch, transport), small_numel=1 << 14,
                       splits={mods[0]: [20000, 50001]})
    grads = []
    for step in range(3):
        g = torch.Generator().manual_seed(1000 * step + rank)
        for m in mods:
            m.params.grad.copy_(torch.randn(m.params.numel(), generator=g) * 10.0 ** float(torch.randint(-4, 0, (1,), generator=g)))
        grads.append([m.params.grad.clone() for m in mods])
        opt.step(lr_scale=0.33 ** (step >= 2))
    shadows = [m.shadow.clone() for m in mods]
    own = [m.params.detach().clone() for m in mods]
    opt.gather_master()
    torch.save({"grads": grads, "shadows": shadows, "own": own, "master": [m.params.detach().clone() for m in mods],
                "wire_bytes": opt.wire_bytes, "ranges": opt.ranges(mods[0])}, os.path.join(out, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _reference_adamw(n_list, seeds, mean_grads, lr_scales):
    ps = [torch.nn.Parameter(_FakeTcnnModule(n, s).params.detach().clone()) for n, s in zip(n_list, seeds)]
    opt = torch.optim.AdamW(ps, lr=0.01, betas=(0.9, 0.99), eps=1e-15)
    for step, gs in enumerate(mean_grads):
        for g_ in opt.param_groups:
            g_["lr"] = 0.01 * lr_scales[step]
        for p, g in zip(ps, gs):
            p.grad = g.clone()
        opt.step()
    return [p.detach() for p in ps]


def test_sharded_adamw_world4_matches_single_process_adamw_on_the_mean_gradient(tmp_path):
    """reduce-scatter -> AdamW on each rank's shard -> all-gather of the fp16 image, world_size 4 over gloo: every rank ends
    with the SAME fp16 image bit for bit, and (fp32 transport) it equals torch.optim.AdamW on the mean gradient"""
    world, port = 4, _free_port()
    mp.spawn(_sharded_worker, args=(world, port, str(tmp_path), "float32"), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"rank{k}.pt") for k in range(world)]
    for k in range(1, world):
        for a, b in zip(r[0]["shadows"], r[k]["shadows"]):
            assert torch.equal(a, b)                                  # replicas stay bit-identical after 3 steps
        for a, b in zip(r[0]["master"], r[k]["master"]):
            assert torch.equal(a, b)                                  # ... and so do the gathered fp32 masters
    mean = [[sum(r[k]["grads"][s][i] for k in range(world)) / world for i in range(2)] for s in range(3)]
    want = _reference_adamw([3072 + 70003, 7168], [7, 8], mean, [1.0, 1.0, 0.33])
    for got, w, sh in zip(r[0]["master"], want, r[0]["shadows"]):
        assert torch.allclose(got, w, rtol=1e-5, atol=1e-7)
        assert torch.equal(sh[:w.numel()], got.half())                # the image the kernels read = rounded master
    # before the gather a rank's fp32 tensor is current in the replicated parts only (MLP head, small module); the
    # table's master values live in the owners' shards
    assert torch.equal(r[1]["own"][0][:3072], r[0]["master"][0][:3072]) and torch.equal(r[1]["own"][1], r[0]["master"][1])
    assert not torch.equal(r[1]["own"][0][3072:], r[0]["master"][0][3072:])
    G = world * 8
    pad = -(-70003 // G) * G
    assert r[0]["ranges"] == [(-(-50001 // G) * G, pad), (-(-20000 // G) * G, -(-50001 // G) * G), (0, -(-20000 // G) * G)]
    n_small = 3072 + 7168
    assert r[0]["wire_bytes"] == pad // world * (world - 1) * (4 + 2) + 2 * n_small * 4 * (world - 1) // world


def test_sharded_adamw_bf16_transport_stays_in_step(tmp_path):
    """the default wire format: bf16 gradients (fp32 range: no scale, no overflow), fp16 image back"""
    world, port = 2, _free_port()
    mp.spawn(_sharded_worker, args=(world, port, str(tmp_path), "bfloat16"), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"rank{k}.pt") for k in range(world)]
    for a, b in zip(r[0]["shadows"], r[1]["shadows"]):
        assert torch.equal(a, b)
    mean = [[sum(r[k]["grads"][s][i] for k in range(world)) / world for i in range(2)] for s in range(3)]
    want = _reference_adamw([3072 + 70003, 7168], [7, 8], mean, [1.0, 1.0, 0.33])
    for got, w in zip(r[0]["master"], want):
        # Adam normalises the step (eps = 1e-15: the first step is lr * sign(g)): a bf16-rounded gradient moves a parameter
        # by ~lr * 2^-8 differently per step -- except where the ranks' gradients cancel to within the rounding and the
        # SIGN of the mean is decided by it (a full lr step, a handful of entries)
        dev = (got - w).abs()
        assert float(dev.mean()) < 1e-4 and float(dev.max()) <= 3 * 0.0101
        assert float((dev > 3 * 0.01 * 2 ** -7).float().mean()) < 5e-3


def _absent_worker(rank, world, port, out):
    for p in (ROOT, os.path.join(ROOT, "instant-nsr-pl_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nsr.parallel import ShardedAdamW
    mods = [_FakeTcnnModule(3072 + 40000, 7, n_network_params=3072)]
    if rank == 1:
        mods[0].params.grad = None  # a rank whose first steps kept no sample never allocated the table gradient
    opt = ShardedAdamW(mods, lr=0.01, transport=torch.float32, small_numel=1 << 10)
    sent = []
    for step in range(3):
        g = torch.Generator().manual_seed(50 * step + rank)
        fresh = torch.randn(3072 + 40000, generator=g) * 1e-2
        kw = {}
        if rank == 1 and step == 0:
            fresh = torch.zeros_like(fresh)          # .grad is None: contributes zeros
        elif rank == 1 and step == 2:
            kw = dict(absent=(mods[0],))              # no backward ran: .grad still holds step 1's gradient
            fresh[3072:] = 0.0                        # (its head is copied from .grad as usual below)
            mods[0].params.grad[:3072].copy_(fresh[:3072])
        if not (rank == 1 and step in (0, 2)):
            if mods[0].params.grad is None:
                mods[0].params.grad = torch.zeros(3072 + 40000)
            mods[0].params.grad.copy_(fresh)
        sent.append(fresh.clone())
        opt.step(overwritten=(mods[0],), **kw)     # the table backward overwrites: nobody clears the body for it
    opt.gather_master()
    torch.save({"sent": sent, "master": mods[0].params.detach().clone()}, os.path.join(out, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_adamw_rank_without_samples_contributes_zeros(tmp_path):
    """a rank whose step launched no table backward (``absent``) or that never allocated the gradient (``.grad`` is
    None) must contribute ZEROS to the reduce-scatter, not the previous step's gradient that ``overwritten`` left in place"""
    world, port = 2, _free_port()
    mp.spawn(_absent_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"rank{k}.pt") for k in range(world)]
    mean = [[(r[0]["sent"][s] + r[1]["sent"][s]) / world] for s in range(3)]
    want = _reference_adamw([3072 + 40000], [7], mean, [1.0, 1.0, 1.0])[0]
    assert torch.equal(r[0]["master"], r[1]["master"])
    assert torch.allclose(r[0]["master"], want, rtol=1e-5, atol=1e-7)


def _grid_sync_worker(rank, world, port, out):
    for p in (ROOT, os.path.join(ROOT, "instant-nsr-pl_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nerfacc import OccupancyGrid
    from nsr.parallel import sync_occupancy_grid
    grid = OccupancyGrid(torch.tensor([-1.0, -1, -1, 1, 1, 1]), resolution=16)
    g = torch.Generator().manual_seed(7 + rank)  # every rank refreshed its own grid from its own random cells
    grid.occs.copy_(torch.rand(grid.occs.shape, generator=g))
    grid._binary.copy_((torch.rand(grid._binary.shape, generator=g) > 0.5))
    before = (grid.occs.clone(), grid._binary.clone())
    sync_occupancy_grid(grid)
    torch.save({"before": before, "after": (grid.occs.clone(), grid._binary.clone())}, os.path.join(out, f"grid{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_occupancy_grids_follow_rank_zero(tmp_path):
    """DDP's broadcast_buffers semantics (reference launch.py:93-107 wraps the system in DDP: rank 0's buffers, the occupancy
    grid's ``occs`` / ``_binary`` among them, are broadcast): after ``sync_occupancy_grid`` every rank holds rank 0's grid"""
    world, port = 2, _free_port()
    mp.spawn(_grid_sync_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, f"grid{r}.pt")) for r in range(world))
    assert not torch.equal(r0["before"][0], r1["before"][0]) and not torch.equal(r0["before"][1], r1["before"][1])
    for k in (0, 1):
        assert torch.equal(r0["after"][k], r0["before"][k])   # rank 0 keeps its grid
        assert torch.equal(r1["after"][k], r0["before"][k])   # ... and rank 1 now has it


class _FakeOpt:
    tcnn_modules = ()

    def state_dict(self):
        return {}

    def load_state_dict(self, sd):
        pass


class _FakeTrainer:
    """just what nsr.trainer.training_state / restore_training_state touch (the sampler state is the subject here)"""
    def __init__(self, rank, world, seed=42):
        from nsr.parallel import shard_seed
        self.rank, self.world_size, self.seed = rank, world, seed
        self.opt, self.sharded, self.global_step, self.train_num_rays = _FakeOpt(), None, 0, 256
        self.gen = torch.Generator()
        self.gen.manual_seed(shard_seed(seed, rank))


def _resume_worker(rank, world, port, out):
    for p in (ROOT, os.path.join(ROOT, "instant-nsr-pl_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nsr.trainer import restore_training_state, training_state
    except Exception:  # the HIP library is needed to import nsr.trainer here: restate the two functions' contract on a stub
        restore_training_state = training_state = None
    res = {"importable": training_state is not None}
    if training_state is not None:
        tr = _FakeTrainer(rank, world)
        for _ in range(3 + rank):
            torch.rand(5, generator=tr.gen)  # ranks are at different points of different streams
        tr.global_step = 20
        st = training_state(tr)  # collective: every rank's sampler state
        expect = torch.rand(8, generator=tr.gen)  # what this rank draws next in the uninterrupted run
        fresh = _FakeTrainer(rank, world)
        restore_training_state(fresh, st)
        res["same_world"] = (expect, torch.rand(8, generator=fresh.gen))
        other = _FakeTrainer(rank, world + 1)  # "resumed at another world size": still distinct streams per rank
        other.world_size = world + 1
        restore_training_state(other, st)
        res["other_world"] = torch.rand(8, generator=other.gen)
        res["n_states"] = len(st["generators"])
    torch.save(res, os.path.join(out, f"resume{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_resume_restores_per_rank_sampler_states(tmp_path):
    """a checkpoint written at world 2 holds BOTH ranks' sampler states; after a resume every rank continues its own
    ray stream (never rank 0's on all ranks), and a resume at another world size still gives the ranks distinct streams"""
    import pytest
    world, port = 2, _free_port()
    mp.spawn(_resume_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, f"resume{r}.pt")) for r in range(world))
    if not r0["importable"]:
        pytest.skip("nsr.trainer needs the built HIP library to import")
    assert r0["n_states"] == r1["n_states"] == 2
    for r in (r0, r1):
        assert torch.equal(r["same_world"][0], r["same_world"][1])  # each rank continues ITS stream
    assert not torch.equal(r0["same_world"][1], r1["same_world"][1])  # ... and they are different streams
    assert not torch.equal(r0["other_world"], r1["other_world"])


def _world8_worker(rank, world, port, out, phase):
    """phase "train": world ranks step 3 times and rank 0 keeps the gathered checkpoint (masters + moments + step count);
    phase "resume": a DIFFERENT world loads it and takes one more step"""
    for p in (ROOT, os.path.join(ROOT, "instant-nsr-pl_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nsr.parallel import ShardedAdamW
    # the real table's raggedness in small: [3072 MLP weights | 12,599,920 mod-like odd body], two level-group cuts
    n_body = 70003
    mods = [_FakeTcnnModule(3072 + n_body, 7, n_network_params=3072), _FakeTcnnModule(7168, 8)]
    opt = ShardedAdamW(mods, lr=0.01, transport=torch.float32, small_numel=1 << 14, splits={mods[0]: [20000, 50001]})
    G = world * 8
    st = opt.state[mods[0]]
    shapes = {"ranges": opt.ranges(mods[0]), "send": int(opt.send_buffer(mods[0]).numel()), "shard": int(st["master"].numel()),
              "shadow": int(st["shadow"].numel())}
    if phase == "resume":
        ck = torch.load(os.path.join(out, "ckpt.pt"))
        for m, p in zip(mods, ck["params"]):
            m.params.data.copy_(p)
        opt.load(mods, ck["moments"], ck["step_count"])
    steps = range(3) if phase == "train" else range(3, 4)
    grads = []
    for step in steps:
        g = torch.Generator().manual_seed(1000 * step + rank)
        for m in mods:
            m.params.grad.copy_(torch.randn(m.params.numel(), generator=g) * 1e-2)
        grads.append([m.params.grad.clone() for m in mods])
        kw = dict(absent=(mods[0],)) if (phase == "train" and step == 1 and rank == world - 1) else {}
        if kw:
            grads[-1][0][3072:] = 0.0  # (no table backward ran on this rank: it contributes zeros for the body)
        opt.step(overwritten=(mods[0],), **kw)
    moments = opt.gather_moments(mods)
    opt.gather_master()
    res = {"grads": grads, "master": [m.params.detach().clone() for m in mods], "shadows": [m.shadow.clone() for m in mods],
           "shapes": shapes, "G": G}
    if phase == "train" and rank == 0:
        torch.save({"params": res["master"], "moments": moments, "step_count": opt.step_count}, os.path.join(out, "ckpt.pt"))
    torch.save(res, os.path.join(out, f"{phase}{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_adamw_world8_shapes_absent_rank_and_resume_at_world2(tmp_path):
    """the shapes `bench.py --gpus 8` meets for the first time on a real node: ranges padded to world x 8 = 64 elements, 1/8
    shards, a rank without samples in one step, the gathered checkpoint of the world-8 run loaded by a world-2 run -- every
    rank bit-identical, the result = torch.optim.AdamW on the mean gradients of all four steps"""
    world, port = 8, _free_port()
    mp.spawn(_world8_worker, args=(world, port, str(tmp_path), "train"), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"train{k}.pt") for k in range(world)]
    G, n_body = 64, 70003
    pad = -(-n_body // G) * G
    sh = r[0]["shapes"]
    assert sh["ranges"] == [(-(-50001 // G) * G, pad), (-(-20000 // G) * G, -(-50001 // G) * G), (0, -(-20000 // G) * G)]
    assert all(a % G == 0 and b % G == 0 for a, b in sh["ranges"])
    assert sh["send"] == pad and sh["shard"] == pad // world and sh["shadow"] >= 3072 + n_body
    for k in range(1, world):
        assert r[k]["shapes"] == sh
        for a, b in zip(r[0]["shadows"], r[k]["shadows"]):
            assert torch.equal(a, b)
        for a, b in zip(r[0]["master"], r[k]["master"]):
            assert torch.equal(a, b)
    mean8 = [[sum(r[k]["grads"][s][i] for k in range(world)) / world for i in range(2)] for s in range(3)]
    # ... resumed by TWO ranks from rank 0's file
    port2 = _free_port()
    mp.spawn(_world8_worker, args=(2, port2, str(tmp_path), "resume"), nprocs=2, join=True)
    q = [torch.load(tmp_path / f"resume{k}.pt") for k in range(2)]
    assert q[0]["shapes"]["shard"] == -(-n_body // 16) * 16 // 2 and q[0]["G"] == 16
    for a, b in zip(q[0]["master"], q[1]["master"]):
        assert torch.equal(a, b)
    mean2 = [[sum(q[k]["grads"][0][i] for k in range(2)) / 2 for i in range(2)]]
    want = _reference_adamw([3072 + n_body, 7168], [7, 8], mean8 + mean2, [1.0, 1.0, 1.0, 1.0])
    for got, w in zip(q[0]["master"], want):
        assert torch.allclose(got, w, rtol=2e-5, atol=2e-7), float((got - w).abs().max())
