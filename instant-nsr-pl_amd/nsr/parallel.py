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
