"""Ray-sharded data parallelism: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" in the CPU tests).  The reference gets this from Lightning DDP (launch.py:93-107); here it is three
small functions so the trainer owns the only collective of the data path -- one mean all-reduce of the gradients.

xGMI is point-to-point (7 links x ~153 GB/s per GPU): the 50 MB hash-table gradient dominates the message, so it
goes FIRST as one un-bucketed all-reduce (RCCL splits it across links itself) and the two tiny MLP gradients follow
as one flattened buffer; there is nothing to overlap it with -- the table gradient is final only when the encoding
backward, the last kernel of the step, has finished.  A step takes ~0.7 ms, a 50 MB ring all-reduce over 8 GPUs ~0.25 ms,
so the large gradients travel as fp16 (scaled by 1024 against underflow: tcnn itself accumulates them in fp16).
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
# Sharded optimizer step: reduce-scatter of the gradient -> AdamW on 1/P of the parameters -> all-gather of
# the fp16 image the kernels read.  Versus "all-reduce, then every rank runs the same dense AdamW":
#   * wire bytes per GPU per step: 2 B (grad, bf16) + 2 B (fp16 shadow) per parameter x (P-1)/P instead of 2 x 4 B x (P-1)/P;
#   * optimizer traffic: 34 B/param on 1/P of the parameters (the dense AdamW sweep is 67 us of a 600 us step at P = 1);
#   * optimizer state (exp_avg, exp_avg_sq) and the fp32 master copy are sharded (ZeRO-1): 12 B/param -> 12/P.
# xGMI on an MI355X node is a full mesh of point-to-point links (7 per GPU), and a ring collective is bound by ONE link.
# Both exchanges are therefore issued as pairwise transfers (all_to_all_single / batched isend-irecv): chunk j goes
# straight to rank j over the dedicated link, all 7 links busy at once (3.1 MB per link at P = 8 for the 12.6 M-parameter
# table instead of 22 MB through a ring).  ``algo="ring"`` keeps RCCL's reduce_scatter_tensor / all_gather_into_tensor for
# comparison.  gloo (CPU tests) has neither all_to_all nor reduce_scatter: the exchanges fall back to all_reduce / all_gather.
# ---------------------------------------------------------------------------------------------------------------------
def _reduce_scatter_mean(send, recv_shard, world, rank, algo):
    """recv_shard[S] (fp32) = mean over ranks of send[rank*S:(rank+1)*S] (send: [P*S], transport dtype)"""
    S = recv_shard.numel()
    backend = dist.get_backend()
    if backend == "gloo":
        tmp = send.float()
        dist.all_reduce(tmp, op=dist.ReduceOp.SUM)
        recv_shard.copy_(tmp[rank * S:(rank + 1) * S] / world)
        return
    if algo == "ring":
        out = torch.empty(S, dtype=send.dtype, device=send.device)
        dist.reduce_scatter_tensor(out, send, op=dist.ReduceOp.SUM)
        recv_shard.copy_(out.float() / world)
        return
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send)  # recv[p*S:(p+1)*S] = rank p's contribution to MY shard
    torch.sum(recv.view(world, S), dim=0, dtype=torch.float32, out=recv_shard)
    recv_shard.div_(world)


def _all_gather_shards(full, shard, world, rank, algo):
    """full[P*S] = concatenation of every rank's shard[S]"""
    S = shard.numel()
    backend = dist.get_backend()
    if backend == "gloo" or algo == "ring":
        if backend == "gloo":  # (gloo gathers CPU tensors only: the 2-ranks-on-one-GPU smoke run goes through the host)
            src = shard.cpu() if shard.is_cuda else shard
            parts = [torch.empty_like(src) for _ in range(world)]
            dist.all_gather(parts, src)
            full.copy_(torch.cat(parts))
        else:
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


class ShardedAdamW:
    """AdamW (torch.optim.AdamW semantics, the reference's optimizer: systems/utils.py:314-325) over the flat fp32
    parameters of tinycudann modules with the gradient exchange folded in (see the block comment above).  Rank r owns
    elements [r S, (r+1) S) of every module's parameter vector: fp32 master values, exp_avg, exp_avg_sq.  After
    ``step`` every rank holds the full, identical fp16 image (``module.adopt_shadow``); the fp32 ``params`` tensor of a rank
    is current only inside its own shard -- ``gather_master()`` completes it (checkpoints)."""

    def __init__(self, modules, lr=0.01, betas=(0.9, 0.99), eps=1e-15, weight_decay=0.01, transport=torch.bfloat16,
                 algo=None):
        assert dist.is_initialized()
        # default: RCCL's own reduce_scatter_tensor / all_gather_into_tensor ("ring" here; RCCL picks its channels over
        # the xGMI mesh itself).  The pairwise variant ("a2a") cannot be exercised on the 1-GPU boxes this was developed
        # on (RCCL refuses two ranks per device), so it is opt-in: NSR_EXCHANGE_ALGO=a2a
        algo = algo or os.environ.get("NSR_EXCHANGE_ALGO", "ring")
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.lr, self.betas, self.eps, self.wd, self.transport, self.algo = lr, betas, eps, weight_decay, transport, algo
        self.step_count = 0
        self.modules = [m for m in modules if m.params.numel() > 0]
        self.state = {}
        self.wire_bytes = 0
        for m in self.modules:
            p = m.params
            n = p.numel()
            S = -(-n // (self.world * 8)) * 8  # shard length: multiple of 8 elements (16-B aligned fp16 rows)
            dev = p.device
            st = dict(n=n, S=S, exp_avg=torch.zeros(S, device=dev), exp_avg_sq=torch.zeros(S, device=dev),
                      grad=torch.zeros(S, device=dev), master=torch.zeros(S, device=dev),
                      send=torch.zeros(S * self.world, dtype=transport, device=dev),
                      shadow_shard=torch.zeros(S, dtype=torch.float16, device=dev),
                      shadow=torch.zeros(S * self.world, dtype=torch.float16, device=dev))
            lo, hi = self.rank * S, min((self.rank + 1) * S, n)
            st["lo"], st["hi"] = lo, max(hi, lo)
            if hi > lo:
                st["master"][:hi - lo].copy_(p.data[lo:hi])
            self.state[m] = st
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            self.wire_bytes += S * (self.world - 1) * (torch.finfo(transport).bits // 8 + 2)

    def _adamw(self, st, lr):
        bc1, bc2 = 1.0 - self.betas[0] ** self.step_count, 1.0 - self.betas[1] ** self.step_count
        p, g, m, v, sh = st["master"], st["grad"], st["exp_avg"], st["exp_avg_sq"], st["shadow_shard"]
        if p.is_cuda:
            from nsr_hip import ops as _ops
            _ops.adamw_step(p, g, m, v, sh, lr, self.betas[0], self.betas[1], self.eps, self.wd, self.step_count,
                            zero_grad=False)
        else:  # gloo tests on CPU tensors: the kernel's arithmetic, in torch
            p.mul_(1.0 - lr * self.wd)
            m.mul_(self.betas[0]).add_(g, alpha=1.0 - self.betas[0])
            v.mul_(self.betas[1]).addcmul_(g, g, value=1.0 - self.betas[1])
            p.addcdiv_(m, v.sqrt() / (bc2 ** 0.5) + self.eps, value=-lr / bc1)
            sh.copy_(p)

    def step(self, lr_scale=1.0, timings=None):
        """timings (dict or None): when given, HIP-event pairs around the two exchanges are appended (bench.py)"""
        self.step_count += 1
        lr = self.lr * lr_scale
        for m in self.modules:
            st, p = self.state[m], m.params
            n, S = st["n"], st["S"]
            ev = None
            if timings is not None and p.is_cuda:
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                ev[0].record()
            st["send"][:n].copy_(p.grad)  # fp32 -> transport dtype (the padding stays zero)
            _reduce_scatter_mean(st["send"], st["grad"], self.world, self.rank, self.algo)
            if ev:
                ev[1].record()
            self._adamw(st, lr)
            if ev:
                ev[2].record()
            _all_gather_shards(st["shadow"], st["shadow_shard"], self.world, self.rank, self.algo)
            if ev:
                ev[3].record()
                timings.setdefault("events", []).append(ev)
            if st["hi"] > st["lo"]:  # keep this rank's slice of the fp32 parameter tensor current
                p.data[st["lo"]:st["hi"]].copy_(st["master"][:st["hi"] - st["lo"]])
            p.grad.zero_()
            if hasattr(m, "adopt_shadow"):
                m.adopt_shadow(st["shadow"][:n])

    def gather_master(self):
        """complete every rank's fp32 ``params`` from the owners' shards (before a checkpoint)"""
        for m in self.modules:
            st = self.state[m]
            full = torch.empty(st["S"] * self.world, device=st["master"].device)
            _all_gather_shards(full, st["master"], self.world, self.rank, "ring")
            m.params.data.copy_(full[:st["n"]])
            if hasattr(m, "invalidate"):
                m.invalidate()
