"""Ray-sharded data parallelism: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" in the CPU tests and in the two-ranks-on-one-GPU smoke runs).  The reference gets this from Lightning DDP
(launch.py:93-107): every rank renders its own ray batch, DDP all-reduces the gradients, every rank runs the same AdamW.

Here the exchange is part of the optimizer step and is shaped for what the path actually moves -- one 12.6 M-parameter
hash table (50 MB as fp32) and ~10 K MLP weights per model:

  * the table gradient is reduce-scattered in **bf16** (fp32 range: no loss scale, no overflow), every rank applies AdamW
    to its 1/P of the table (fp32 master values and moments sharded: ZeRO-1), and the **fp16 image** the kernels read is
    all-gathered straight into the tensor they read it from.  Wire bytes per GPU and step: 2 x 2 B x (P-1)/P per parameter
    instead of 2 x 4 B for an fp32 ring all-reduce, and the dense optimizer sweep shrinks to 1/P;
  * the table is exchanged in RANGES that become final one after the other: the fused step launches its table backward
    finest levels first (csrc/step.hip nsr_nerf_main_pass_exchange), records an event behind each level group, and the
    exchange -- issued on a communication stream -- reduce-scatters range k while the backward still accumulates range
    k + 1.  The backward kernel writes bf16 straight into the send buffer: no fp32 gradient, no cast pass, no memset;
  * everything small (MLP weights in front of a table, the colour MLP) travels as ONE flattened fp32 all-reduce that is
    issued as soon as the weight-gradient kernels are done -- underneath the table backward -- and is stepped by every
    rank redundantly (a 10 K-element AdamW is cheaper than a second collective).

xGMI is point-to-point (7 links x ~153 GB/s per GPU), so a collective costs a latency floor of tens of microseconds plus
its bytes over the links: few, large collectives.  Default = RCCL's own reduce_scatter_tensor / all_gather_into_tensor
(``algo="ring"``; RCCL maps its channels onto the mesh itself); ``NSR_EXCHANGE_ALGO=a2a`` issues them as pairwise transfers
(all_to_all_single / batched isend-irecv: chunk j straight to rank j over its dedicated link).
"""
import os

import torch
import torch.distributed as dist


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_seed(seed, rank):
    """Per-rank sampling seed.  The reference seeds every DDP rank identically (launch.py:62-64), so its ranks draw
    the SAME ray batch; sharding rays needs distinct streams."""
    return int(seed) + 1000003 * int(rank)


def broadcast_parameters(module, src=0):
    """replicas start identical (DDP does this at construction)"""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        if t.numel() > 0:
            dist.broadcast(t.data, src=src)
    for m in module.modules():  # .data writes do not bump the version the fp16 shadow of a tcnn module is keyed on
        if hasattr(m, "invalidate"):
            m.invalidate()


def sync_occupancy_grid(grid, src=0):
    """every rank continues with rank ``src``'s occupancy grid.  The reference gets this from DDP (``broadcast_buffers=True``,
    launch.py:93-107: rank 0's buffers -- the grid's ``occs`` / ``_binary`` among them -- are broadcast at every forward); the
    ranks refresh their grids from identical weights, but the refresh draws random cells and jitters, and grids that drift
    apart would march different sample sets through replicas that are supposed to see the same model.  Called behind every
    refresh (each 16th step: 10 MB over xGMI); re-packs the 4^3-brick bitfield the marchers read."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    binary = grid._binary
    u8 = binary.view(torch.uint8) if binary.dtype == torch.bool else binary
    dist.broadcast(grid.occs, src=src)
    dist.broadcast(u8, src=src)
    tag = getattr(binary, "_nsr_bricks", None)
    if tag is not None and binary.is_cuda:  # the packed copy (a persistent buffer other code holds on to): re-pack in place
        from nsr_hip import check, lib, ptr, stream_ptr
        rx, ry, rz = (int(v) for v in binary.shape)
        with torch.cuda.device(binary.device):
            check(lib.nsr_grid_pack_bricks(ptr(u8), rx, ry, rz, ptr(tag[2]), stream_ptr()), "nsr_grid_pack_bricks")


HALF_TRANSPORT_SCALE = 1024.0


def _all_reduce_half(g, world):
    """g <- mean over ranks, summed in fp16 on the wire"""
    n = g.numel()
    h = torch.empty(n, dtype=torch.float16, device=g.device)
    if g.is_cuda:
        from nsr_hip import check, lib, ptr, stream_ptr
        with torch.cuda.device(g.device):
            check(lib.nsr_scale_to_half(ptr(g), ptr(h), n, HALF_TRANSPORT_SCALE, stream_ptr()), "nsr_scale_to_half")
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            check(lib.nsr_scale_from_half(ptr(h), ptr(g), n, 1.0 / (HALF_TRANSPORT_SCALE * world), stream_ptr()),
                  "nsr_scale_from_half")
    else:  # gloo tests on CPU tensors
        h.copy_(g.reshape(-1) * HALF_TRANSPORT_SCALE)
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        g.copy_((h.float() * (1.0 / (HALF_TRANSPORT_SCALE * world))).view_as(g))


def all_reduce_gradients(params, small_numel=1 << 16, half_transport=False):
    """mean all-reduce of ``.grad`` over all ranks: large tensors individually (largest first; as fp16 on the wire
    with ``half_transport``), small ones flattened into one fp32 message."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    world = dist.get_world_size()
    grads = [p.grad for p in params if p.grad is not None and p.grad.numel() > 0]
    big = sorted([g for g in grads if g.numel() > small_numel], key=lambda g: -g.numel())
    small = [g for g in grads if g.numel() <= small_numel]
    n_bytes = 0
    for g in big:
        if half_transport and g.dtype == torch.float32 and g.is_contiguous():
            _all_reduce_half(g, world)
            n_bytes += g.numel() * 2
        else:
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
            g.div_(world)
            n_bytes += g.numel() * g.element_size()
    if small:
        flat = torch.cat([g.reshape(-1) for g in small])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(world)
        off = 0
        for g in small:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        n_bytes += flat.numel() * flat.element_size()
    return n_bytes


# ---------------------------------------------------------------------------------------------------------------------
# the two exchanges of a range.  gloo (CPU tests, two ranks on one GPU) has neither reduce_scatter nor all_to_all: there
# they fall back to all_reduce / all_gather of host tensors -- same result, the logic around them is what those runs cover
# ---------------------------------------------------------------------------------------------------------------------
def _reduce_scatter_sum(send, recv_shard, world, rank, algo):
    """recv_shard[S] (transport dtype) = SUM over ranks of send[rank*S:(rank+1)*S]  (send: [P*S])"""
    S = recv_shard.numel()
    backend = dist.get_backend()
    if backend == "gloo":
        tmp = send.float().cpu() if send.is_cuda else send.float()
        dist.all_reduce(tmp, op=dist.ReduceOp.SUM)
        recv_shard.copy_(tmp[rank * S:(rank + 1) * S])
        return
    if algo == "ring":
        dist.reduce_scatter_tensor(recv_shard, send, op=dist.ReduceOp.SUM)
        return
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send)  # recv[p*S:(p+1)*S] = rank p's contribution to MY shard
    recv_shard.copy_(torch.sum(recv.view(world, S), dim=0, dtype=torch.float32))


def _all_gather_shards(full, shard, world, rank, algo):
    """full[P*S] = concatenation of every rank's shard[S]"""
    S = shard.numel()
    backend = dist.get_backend()
    if backend == "gloo":
        src = shard.cpu() if shard.is_cuda else shard
        parts = [torch.empty_like(src) for _ in range(world)]
        dist.all_gather(parts, src)
        full.copy_(torch.cat(parts))
        return
    if algo == "ring":
        dist.all_gather_into_tensor(full, shard)
        return
    full[rank * S:(rank + 1) * S].copy_(shard)
    ops = []
    for p in range(world):
        if p != rank:
            ops.append(dist.P2POp(dist.isend, shard, p))
            ops.append(dist.P2POp(dist.irecv, full[p * S:(p + 1) * S], p))
    for req in dist.batch_isend_irecv(ops):
        req.wait()


def _round_up(v, g):
    return -(-int(v) // g) * g


class ShardedAdamW:
    """AdamW (torch.optim.AdamW semantics, the reference's optimizer: systems/utils.py:314-325) over the flat fp32
    parameters of tinycudann modules with the gradient exchange folded in (see the module docstring).

    A module's parameter vector is ``[head | body]``: the head (the MLP weights in front of a hash table; a module of at
    most ``small_numel`` parameters is all head) is all-reduced and stepped by every rank, the body (the table) is cut into
    ranges at ``splits[module]`` (body-relative element offsets, e.g. level boundaries; rounded up to P x 8 elements) and
    each range is reduce-scattered, stepped on the owner's 1/P and all-gathered as fp16.  Ranges are exchanged from the
    highest offsets down (the fused step finishes the finest levels first).

    After ``step`` every rank holds the full, identical fp16 image (``module.adopt_shadow``).  The fp32 ``params`` tensor of a
    module is current in its head only: the body's master values live in the owners' shards until ``gather_master()``
    completes them on every rank (checkpoints: ``Trainer.state_dict``)."""

    def __init__(self, modules, lr=0.01, betas=(0.9, 0.99), eps=1e-15, weight_decay=0.01, transport=torch.bfloat16,
                 algo=None, small_numel=1 << 16, splits=None):
        assert dist.is_initialized()
        algo = algo or os.environ.get("NSR_EXCHANGE_ALGO", "ring")
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.lr, self.betas, self.eps, self.wd, self.transport, self.algo = lr, betas, eps, weight_decay, transport, algo
        self.step_count = 0
        self.modules = [m for m in modules if m.params.numel() > 0]
        self.state, self.small, self.wire_bytes = {}, [], 0
        self.master_current = True  # False after a step until gather_master(): the fp32 bodies are then stale
        P, G = self.world, self.world * 8  # shards are multiples of 8 elements: 16-byte aligned fp16 rows
        splits = splits or {}
        n_small = 0
        for m in self.modules:
            p = m.params
            n, dev = p.numel(), p.device
            head = n if n <= small_numel else int(getattr(m, "n_network_params", 0) or 0)
            body = n - head
            if body and head % 8:
                raise NotImplementedError("ShardedAdamW: the head in front of a table must be a multiple of 8 elements")
            body_pad = _round_up(body, G) if body else 0
            st = dict(n=n, head=head, body=body, body_pad=body_pad, small_off=n_small)
            n_small += _round_up(head, 8)  # (16-byte aligned pieces: the AdamW kernel moves float4s)
            if p.grad is None and body == 0:
                p.grad = torch.zeros_like(p)
            # the fp16 image of the WHOLE vector (padded): the all-gather writes into it, the kernels read it
            st["shadow"] = torch.zeros(head + body_pad, dtype=torch.float16, device=dev)
            st["shadow"][:n].copy_(p.data)
            if hasattr(m, "adopt_shadow"):
                m.adopt_shadow(st["shadow"][:n])
            if body:
                cuts = sorted({min(_round_up(c, G), body_pad) for c in splits.get(m, ()) if 0 < c < body})
                edges = [0] + [c for c in cuts if 0 < c < body_pad] + [body_pad]
                st["ranges"] = [(a, b) for a, b in zip(edges[:-1], edges[1:])][::-1]  # highest offsets first
                S_all = body_pad // P
                st["send"] = torch.zeros(body_pad, dtype=transport, device=dev)
                st["recv"] = torch.zeros(S_all, dtype=transport, device=dev)
                for k in ("master", "exp_avg", "exp_avg_sq", "g32"):
                    st[k] = torch.zeros(S_all, device=dev)
                st["shard16"] = torch.zeros(S_all, dtype=torch.float16, device=dev)
                self._scatter_body(st, "master", p.data)
                self.wire_bytes += S_all * (P - 1) * (torch.finfo(transport).bits // 8 + 2)
            else:
                st["ranges"] = []
            self.state[m] = st
        dev = self.modules[0].params.device if self.modules else None
        self.small_grad = torch.zeros(n_small, device=dev) if n_small else None
        self.small_m = torch.zeros(n_small, device=dev) if n_small else None
        self.small_v = torch.zeros(n_small, device=dev) if n_small else None
        if n_small:
            self.wire_bytes += 2 * n_small * 4 * (P - 1) // P
        self._comm, self._done, self._done_timed = None, None, None

    def _scatter_body(self, st, key, full):
        """st[key] (this rank's shard, range by range) <- the body part of the full vector ``full`` [n]"""
        P, head, body = self.world, st["head"], st["body"]
        for a, b in st["ranges"]:  # this rank's slice of every range: [a + r S, a + (r + 1) S), S = (b - a) / P
            S = (b - a) // P
            lo, hi = a + self.rank * S, min(a + (self.rank + 1) * S, body)
            if hi > lo:
                st[key][a // P:a // P + hi - lo].copy_(full[head + lo:head + hi])

    def _gather_body(self, st, key, out):
        """out [n]: body part <- every rank's shard st[key] (a collective)"""
        full = torch.empty(st["body_pad"], device=st[key].device)
        for a, b in st["ranges"]:
            S = (b - a) // self.world
            _all_gather_shards(full[a:b], st[key][a // self.world:a // self.world + S].contiguous(), self.world, self.rank, "ring")
        out[st["head"]:].copy_(full[:st["body"]])

    # ---- checkpoints -------------------------------------------------------------------------------------------------------
    def gather_moments(self, modules):
        """[(exp_avg, exp_avg_sq)] as FULL fp32 vectors, one pair per module of ``modules`` (a collective: every rank calls
        it): the checkpoint format is the single-process optimizer's, so a run resumes at any world size"""
        out = []
        for m in modules:
            st = self.state[m]
            ea, eas = torch.zeros(st["n"], device=m.params.device), torch.zeros(st["n"], device=m.params.device)
            o, h = st["small_off"], st["head"]
            if h:
                ea[:h].copy_(self.small_m[o:o + h])
                eas[:h].copy_(self.small_v[o:o + h])
            if st["ranges"]:
                self._gather_body(st, "exp_avg", ea)
                self._gather_body(st, "exp_avg_sq", eas)
            out.append((ea, eas))
        return out

    def load(self, modules, moments, step_count):
        """re-seed from the modules' (freshly loaded) fp32 ``params``: sharded masters, fp16 shards and the fp16 image; with
        ``moments`` ([(exp_avg, exp_avg_sq)] full vectors, ``gather_moments`` order) also the optimizer state"""
        for k, m in enumerate(modules):
            st = self.state[m]
            p = m.params
            st["shadow"][:st["n"]].copy_(p.data)
            if st["ranges"]:
                self._scatter_body(st, "master", p.data)
                st["shard16"].copy_(st["master"])
            if moments is not None:
                ea, eas = moments[k]
                o, h = st["small_off"], st["head"]
                if h:
                    self.small_m[o:o + h].copy_(ea[:h])
                    self.small_v[o:o + h].copy_(eas[:h])
                if st["ranges"]:
                    self._scatter_body(st, "exp_avg", ea)
                    self._scatter_body(st, "exp_avg_sq", eas)
            if hasattr(m, "adopt_shadow"):
                m.adopt_shadow(st["shadow"][:st["n"]])
        if step_count is not None:
            self.step_count = int(step_count)
        self.master_current = True

    # ---- what a fused step writes into directly ---------------------------------------------------------------------
    def send_buffer(self, module):
        """the body's gradient in transport format ([body_pad], element 0 = first body parameter): a step that fills it
        itself (the table backward writing bf16) passes the module in ``step(prefilled=...)``"""
        return self.state[module]["send"]

    def small_grad_view(self, module):
        """fp32 gradient of the module's head inside the flattened small-gradient message (zeroed by ``step``)"""
        st = self.state[module]
        return self.small_grad[st["small_off"]:st["small_off"] + st["head"]]

    def ranges(self, module):
        """body-relative element ranges in the order they are exchanged"""
        return list(self.state[module]["ranges"])

    # ---- AdamW on a shard / on the replicated small tensors -----------------------------------------------------------
    def _adamw(self, p, g, m, v, sh, lr, grad_unscale):
        if p.is_cuda:
            from nsr_hip import ops as _ops
            _ops.adamw_step(p, g, m, v, sh, lr, self.betas[0], self.betas[1], self.eps, self.wd, self.step_count,
                            grad_unscale=grad_unscale, zero_grad=True)
        else:  # gloo tests on CPU tensors: the kernel's arithmetic, in torch
            bc1, bc2 = 1.0 - self.betas[0] ** self.step_count, 1.0 - self.betas[1] ** self.step_count
            gg = g * grad_unscale
            p.mul_(1.0 - lr * self.wd)
            m.mul_(self.betas[0]).add_(gg, alpha=1.0 - self.betas[0])
            v.mul_(self.betas[1]).addcmul_(gg, gg, value=1.0 - self.betas[1])
            p.addcdiv_(m, v.sqrt() / (bc2 ** 0.5) + self.eps, value=-lr / bc1)
            sh.copy_(p)
            g.zero_()

    def step(self, lr_scale=1.0, timings=None, ready=None, prefilled=(), direct_small=(), overwritten=(), absent=()):
        """One optimizer step over all modules, exchange included.

        ``prefilled``: modules whose send buffer already holds this step's body gradient (otherwise it is cast from
        ``params.grad``).  ``direct_small``: modules whose head gradient was written into ``small_grad_view`` (otherwise it
        is copied from ``params.grad`` and that slice is re-zeroed).  ``overwritten``: modules whose body gradient in
        ``params.grad`` is overwritten by every step (the owner-computes table backward) -- the others' is re-zeroed after the
        cast.  ``absent``: modules whose backward did not run this step (a rank that kept no sample): their body
        contributes zeros -- whatever ``params.grad`` holds is a previous step's gradient.  A module whose ``params.grad`` is
        None contributes zeros too.  ``ready``: {"small": event, module: [event per range, in
        exchange order]} -- torch events the communication stream waits for before it touches the respective gradient
        (default: everything queued on the current stream so far).  ``timings`` (dict or None): when given, HIP-event
        tuples around reduce-scatter / AdamW / all-gather of every range are appended to ``timings["events"]`` and around
        the small all-reduce to ``timings["small"]`` (bench.py)."""
        self.step_count += 1
        lr = self.lr * lr_scale
        inv_world = 1.0 / self.world
        ready = ready or {}
        cuda = bool(self.modules) and self.modules[0].params.is_cuda
        overlap = cuda and dist.get_backend() != "gloo"  # gloo collectives block the host: nothing to overlap
        main = torch.cuda.current_stream() if cuda else None
        if overlap:
            if self._comm is None:
                from nsr_hip import shared_stream as _shared_stream
                self._comm = _shared_stream(self.modules[0].params.device, "comm")
            comm = self._comm
            now = torch.cuda.Event()
            now.record(main)
        tev = (lambda: torch.cuda.Event(enable_timing=True)) if (timings is not None and cuda) else None

        def wait_for(ev):
            if overlap:
                comm.wait_event(ev if ev is not None else now)

        ctx = torch.cuda.stream(comm) if overlap else _NullCtx()
        with ctx:
            # 1. everything small: one flattened fp32 all-reduce, stepped redundantly on every rank
            if self.small_grad is not None:
                from_grad = [m for m in self.modules if self.state[m]["head"] and m not in direct_small]
                wait_for(None if from_grad else ready.get("small"))
                for m in self.modules:
                    st = self.state[m]
                    if st["head"] and m not in direct_small:
                        g = m.params.grad
                        view = self.small_grad_view(m)
                        if g is not None and g.data_ptr() != view.data_ptr():
                            view.copy_(g[:st["head"]])
                            g[:st["head"]].zero_()
                e = [tev(), tev()] if tev else None
                if e:
                    e[0].record()
                dist.all_reduce(self.small_grad, op=dist.ReduceOp.SUM)
                if e:
                    e[1].record()
                    timings.setdefault("small", []).append(e)
                for m in self.modules:
                    st = self.state[m]
                    if st["head"]:
                        o, h = st["small_off"], st["head"]
                        self._adamw(m.params.data[:h], self.small_grad[o:o + h], self.small_m[o:o + h],
                                    self.small_v[o:o + h], st["shadow"][:h], lr, inv_world)
            # 2. the tables, range by range: reduce-scatter (bf16) -> AdamW on this rank's shard -> all-gather (fp16 image)
            for m in self.modules:
                st = self.state[m]
                evs = ready.get(m)
                for k, (a, b) in enumerate(st["ranges"]):
                    wait_for(evs[k] if evs else None)
                    head, S = st["head"], (b - a) // self.world
                    if m not in prefilled:
                        hi = min(b, st["body"])
                        g = m.params.grad
                        if g is None or m in absent:
                            st["send"][a:hi].zero_()
                        else:
                            st["send"][a:hi].copy_(g[head + a:head + hi])  # fp32 -> transport dtype
                            if m not in overwritten:
                                g[head + a:head + hi].zero_()
                    sl = slice(a // self.world, a // self.world + S)
                    e = [tev() for _ in range(4)] if tev else None
                    if e:
                        e[0].record()
                    _reduce_scatter_sum(st["send"][a:b], st["recv"][sl], self.world, self.rank, self.algo)
                    if e:
                        e[1].record()
                    st["g32"][sl].copy_(st["recv"][sl])
                    self._adamw(st["master"][sl], st["g32"][sl], st["exp_avg"][sl], st["exp_avg_sq"][sl], st["shard16"][sl],
                                lr, inv_world)
                    if e:
                        e[2].record()
                    _all_gather_shards(st["shadow"][head + a:head + b], st["shard16"][sl], self.world, self.rank, self.algo)
                    if e:
                        e[3].record()
                        timings.setdefault("events", []).append(e)
                if hasattr(m, "adopt_shadow"):
                    m.adopt_shadow(st["shadow"][:st["n"]])
            if tev:  # (bench.py: how long the step's own stream has to wait for the exchange)
                self._done_timed = tev()
                self._done_timed.record()
            if overlap:
                self._done = torch.cuda.Event()
                self._done.record(comm)
        if overlap:
            main.wait_event(self._done)  # the next step's first kernel reads the gathered image
        self.master_current = not any(self.state[m]["ranges"] for m in self.modules)

    def gather_master(self):
        """complete every rank's fp32 ``params`` from the owners' shards (before a checkpoint; a collective: every rank
        calls it)"""
        for m in self.modules:
            st = self.state[m]
            if not st["ranges"]:
                continue
            self._gather_body(st, "master", m.params.data)
        self.master_current = True


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
