"""One-process-per-GPU training step around the hot path (mirror of reference ``systems/nerf.py:87-122``,
``systems/base.py:54-57``): sample rays -> refresh occupancy -> render -> loss -> backward -> all-reduce -> AdamW.

Multi-GPU: rays shard naturally.  Every rank draws its OWN ray batch (rank-offset RNG: the reference seeds all ranks
identically, launch.py:62-64, so its DDP ranks render duplicate batches), replicates the 50 MB model, and the only
data-path collective is one mean all-reduce of the gradients per step over RCCL (``torch.distributed`` backend "nccl").

``Trainer(async_mode=True)`` is the measured path: no host synchronisation inside a step (device-side sample and ray
counts, lagged capacity control), the marching passes two steps ahead on a side stream, the occupancy refresh on the
device.  ``async_mode=False`` keeps the step that reads its counts back (and ``fused=False`` the modular autograd path).
"""
import ctypes
import os

import torch
import torch.distributed as dist
import torch.nn.functional as F

import tinycudann as tcnn
from nsr_hip import ops as _ops
from nsr_hip import check as _check, lib as _lib, ptr as _ptr, shared_stream as _shared_stream, stream_ptr as _stream_ptr

from .parallel import ShardedAdamW, all_reduce_gradients, broadcast_parameters, shard_seed, sync_occupancy_grid


class FusedAdamW:
    """AdamW over the flat fp32 parameters with ONE kernel per tensor that also unscales, refreshes the fp16 shadow
    the kernels read and zeroes the gradient (configs/nerf-blender.yaml:74-79: lr 0.01, betas (0.9,0.99), eps 1e-15)."""

    def __init__(self, named_modules, other_params, lr=0.01, betas=(0.9, 0.99), eps=1e-15, weight_decay=0.01):
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.step_count = 0
        self.table_grad_overwritten = False  # set by the fused trainers: their table backward writes every entry
        self.tcnn_modules = [m for m in named_modules if isinstance(m, tcnn.Module) and m.params.numel() > 0]
        self.state = {}
        for m in self.tcnn_modules:
            p = m.params
            self.state[p] = (torch.zeros_like(p), torch.zeros_like(p), torch.empty_like(p, dtype=torch.float16))
            p.grad = torch.zeros_like(p)
        self.other = torch.optim.AdamW(other_params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay) \
            if other_params else None

    def step(self, lr_scale=1.0, grad_unscale=1.0, updated_in_backward=()):
        """``updated_in_backward``: modules whose (whole) parameter vector was already stepped inside their table backward
        (table_update_desc with this step's lr): only their fp16 image is adopted, and the device-side step counter /
        running beta powers those kernels read are advanced"""
        self.step_count += 1
        if updated_in_backward or self._step_dev is not None:
            # (once the device-side counter exists it advances with EVERY step -- also one whose backward launched nothing
            # and whose tables therefore take the sweep below -- or its bias corrections would fall behind the host's)
            step_dev, hyper = self._device_schedule_state()
            _ops.adam_tick(step_dev, hyper, self.lr * lr_scale, self.betas[0], self.betas[1], 1.0, ())
        for m in self.tcnn_modules:
            p = m.params
            exp_avg, exp_avg_sq, shadow = self.state[p]
            if any(m is u for u in updated_in_backward):
                m.adopt_shadow(shadow)
                continue
            _ops.adamw_step(p.data, p.grad, exp_avg, exp_avg_sq, shadow, self.lr * lr_scale, self.betas[0],
                            self.betas[1], self.eps, self.wd, self.step_count, grad_unscale=grad_unscale, zero_grad=True)
            # the kernel already wrote the fp16 copy: hand it to the module instead of re-casting 12.6 M floats
            m.adopt_shadow(shadow)
        if self.other is not None:
            for g in self.other.param_groups:
                g["lr"] = self.lr * lr_scale
            self.other.step()
            self.other.zero_grad(set_to_none=False)

    # ---- checkpoints (reference launch.py --resume: Lightning restores optimizer state and step count) ----------------------
    def state_dict(self):
        return {"step_count": self.step_count,
                "moments": [(self.state[m.params][0].detach().clone(), self.state[m.params][1].detach().clone())
                            for m in self.tcnn_modules],
                "other": self.other.state_dict() if self.other is not None else None}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step_count"])
        for m, (ea, eas) in zip(self.tcnn_modules, sd["moments"]):
            self.state[m.params][0].copy_(ea)
            self.state[m.params][1].copy_(eas)
        if self.other is not None and sd.get("other") is not None:
            self.other.load_state_dict(sd["other"])
        self.resync()

    def resync(self):
        """after the parameters were written from outside (``load_state_dict``): the fp16 images the kernels read are
        re-cast from them and the device-side schedule state (step counter, running beta powers) restarts from the host's
        step count"""
        for m in self.tcnn_modules:
            shadow = self.state[m.params][2]
            shadow.copy_(m.params.data)
            m.adopt_shadow(shadow)
        if self._step_dev is not None:
            self._step_bufs.fill_(self.step_count)
            self._hyper_bufs.zero_()  # (power cache keyed on a step that no longer matches: recomputed by the next launch)

    # the device-side schedule state is a DOUBLE buffer: ``step_device(other_stream_reads=True)`` writes the advanced state
    # into the other half and flips, so that the table backward's fused AdamW (main stream) may read this step's half while
    # the MLP launch that advances the schedule runs on the helper stream
    @property
    def _step_dev(self):
        return None if getattr(self, "_step_bufs", None) is None else self._step_bufs[self._cur]

    @property
    def _hyper(self):
        return None if getattr(self, "_step_bufs", None) is None else self._hyper_bufs[self._cur]

    def _device_schedule_state(self):
        dev = self.tcnn_modules[0].params.device
        if getattr(self, "_step_bufs", None) is None:
            self._cur = 0
            self._step_bufs = torch.full((2, 4), self.step_count, dtype=torch.int32, device=dev)[:, :1]
            self._hyper_bufs = torch.zeros(2, 16, dtype=torch.float32, device=dev)[:, :12]  # lr, bc1, bc2 | running beta powers | ticket
        return self._step_dev, self._hyper

    def table_update_desc(self, module, milestones=(10000, 15000, 18000), gamma=0.33, lr=None):
        """``NsrTableAdam`` for ``module`` (a NetworkWithInputEncoding): AdamW on its hash table applied INSIDE the table
        backward of the asynchronous step (csrc/hashgrid.hip OwnerAdam); ``step_device(skip_table_of=module)`` then
        updates what is left (the MLP weights) and advances the device-side schedule."""
        from nsr_hip import NsrTableAdam
        step_dev, hyper = self._device_schedule_state()
        p = module.params
        exp_avg, exp_avg_sq, shadow = self.state[p]
        # (built once per half of the schedule's double buffer: a ctypes struct per step is host time the asynchronous step
        # does not have -- its host and GPU times are within 15 % of each other)
        ck = (id(module), self._cur, tuple(milestones), gamma, lr, p.data_ptr(), exp_avg.data_ptr(), shadow.data_ptr())
        cache = self.__dict__.setdefault("_table_desc_cache", {})
        if ck in cache:
            return cache[ck]
        if len(cache) > 8:
            cache.clear()
        n0 = int(getattr(module, "n_network_params", 0))  # (a bare tcnn.Encoding: the table is the whole vector)
        ms = [int(m) for m in milestones][:3] + [0x7fffffff] * (3 - min(len(milestones), 3))
        d = NsrTableAdam()
        d.params, d.exp_avg, d.exp_avg_sq = (t.data_ptr() + 4 * n0 for t in (p.data, exp_avg, exp_avg_sq))
        d.shadow = shadow.data_ptr() + 2 * n0
        d.step, d.hyper = step_dev.data_ptr(), hyper.data_ptr()
        # ``lr``: this step's learning rate from a host-side schedule (then without milestones / gamma 1)
        d.base_lr = float(self.lr if lr is None else lr)
        d.beta1, d.beta2, d.gamma = float(self.betas[0]), float(self.betas[1]), float(gamma)
        d.milestone0, d.milestone1, d.milestone2 = ms
        d.eps, d.weight_decay = float(self.eps), float(self.wd)
        cache[ck] = d
        return d

    def step_device(self, milestones=(10000, 15000, 18000), gamma=0.33, skip_table_of=None, other_stream_reads=False,
                    stream=None):
        """the same update with the step counter, MultiStepLR scale and bias corrections kept ON THE DEVICE
        (nsr_adam_tick): no per-step host scalar, so the launches can be replayed from a captured graph"""
        if self.other is not None:
            raise NotImplementedError("step_device covers models whose parameters all live in fused tcnn modules")
        self._device_schedule_state()
        self.step_count += 1  # host mirror (not read by the kernels)
        if skip_table_of is not None:
            # the table of ``skip_table_of`` was updated inside its backward (table_update_desc): what is left are the MLP
            # weights in front of it and the other module -- one launch, which also advances the schedule
            m0, n0 = skip_table_of, int(skip_table_of.n_network_params)
            if other_stream_reads and stream is not None and m0.params.grad is not None:
                # the asynchronous step's launch, with its arguments resolved once per half of the schedule's double buffer
                # (every pointer and scalar that is baked into the cached argument list is part of the key: ADVICE r5)
                others = [m for m in self.tcnn_modules if m is not m0]
                ck = (id(m0), self._cur, tuple(milestones), gamma, float(self.lr), float(self.eps), float(self.wd),
                      tuple(self.betas)) + tuple(
                    t.data_ptr() for m in [m0] + others
                    for t in (m.params.data, m.params.grad) + tuple(self.state[m.params]) if t is not None)
                cache = self.__dict__.setdefault("_step_args_cache", {})
                args = cache.get(ck)
                if args is None:
                    if len(cache) > 8:
                        cache.clear()
                    rest = [m for m in self.tcnn_modules if m is not m0]
                    assert len(rest) <= 1 and n0 > 0 and n0 % 4 == 0 and all(m.params.grad is not None for m in rest)
                    P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
                    a = (m0.params.data, m0.params.grad) + tuple(self.state[m0.params])
                    ms = [int(m) for m in milestones][:3] + [0x7fffffff] * (3 - min(len(milestones), 3))
                    if rest:
                        b = (rest[0].params.data, rest[0].params.grad) + tuple(self.state[rest[0].params])
                        bargs = tuple(P(t) for t in b) + (b[0].numel(),)
                    else:
                        bargs = (None,) * 5 + (0,)
                    args = cache[ck] = (tuple(P(t) for t in a) + (n0, 0) + bargs +
                                        (P(self._step_bufs[self._cur]), P(self._hyper_bufs[self._cur]),
                                         P(self._step_bufs[self._cur ^ 1]), P(self._hyper_bufs[self._cur ^ 1]),
                                         float(self.lr), float(self.betas[0]), float(self.betas[1]), float(gamma), ms[0], ms[1], ms[2],
                                         float(self.eps), float(self.wd), 1.0, 1), rest)
                _check(_lib.nsr_adamw_step_scheduled_to(*args[0], stream), "nsr_adamw_step_scheduled")
                self._cur ^= 1
                for m in self.tcnn_modules:
                    m.adopt_shadow(self.state[m.params][2])
                return
            rest = [m for m in self.tcnn_modules if m is not m0]
            assert len(rest) <= 1 and n0 > 0 and n0 % 4 == 0
            segs = [tuple(t[:n0] for t in (m0.params.data, m0.params.grad) + tuple(self.state[m0.params])) + (0,)]
            segs += [(m.params.data, m.params.grad) + tuple(self.state[m.params]) + (0,) for m in rest]
            out = None
            if other_stream_reads:
                out = (self._step_bufs[self._cur ^ 1], self._hyper_bufs[self._cur ^ 1])
            _ops.adamw_step_scheduled(segs, self._step_dev, self._hyper, self.lr, self.betas[0], self.betas[1], gamma,
                                      milestones, self.eps, self.wd, out=out, stream=stream)
            if other_stream_reads:
                self._cur ^= 1
            for m in self.tcnn_modules:
                m.adopt_shadow(self.state[m.params][2])
            return
        assert not other_stream_reads

        def zero_n(m):  # the fused step overwrites the hash-table slice of the gradient: zero only the MLP slice in front
            n_zero = getattr(m, "n_network_params", 0) if getattr(m, "grid_desc", None) is not None else 0
            return n_zero if (n_zero > 0 and n_zero % 4 == 0 and self.table_grad_overwritten) else 0

        mods = self.tcnn_modules
        if len(mods) <= 2 and not os.environ.get("NSR_ADAM_SEPARATE"):  # schedule tick + every tensor in ONE launch (largest tensor first: it sizes the grid)
            mods = sorted(mods, key=lambda m: -m.params.numel())
            _ops.adamw_step_scheduled([(m.params.data, m.params.grad) + tuple(self.state[m.params]) + (zero_n(m),)
                                       for m in mods], self._step_dev, self._hyper, self.lr, self.betas[0],
                                      self.betas[1], gamma, milestones, self.eps, self.wd)
        else:
            _ops.adam_tick(self._step_dev, self._hyper, self.lr, self.betas[0], self.betas[1], gamma, milestones)
            for m in mods:
                p = m.params
                exp_avg, exp_avg_sq, shadow = self.state[p]
                _ops.adamw_step(p.data, p.grad, exp_avg, exp_avg_sq, shadow, self.lr, self.betas[0], self.betas[1],
                                self.eps, self.wd, self.step_count, zero_grad=True, hyper=self._hyper,
                                zero_first_n=zero_n(m))
        for m in mods:
            m.adopt_shadow(self.state[m.params][2])


def checkpoint_state_dict(model, sharded):
    if sharded is not None:
        sharded.gather_master()
    return model.state_dict()


def training_state(trainer, extra_optimizers=()):
    """everything a resumed run needs beside the model (reference launch.py --resume / Lightning's checkpoint): optimizer
    moments and step counts (gathered from the ranks' shards at world > 1: a COLLECTIVE, and the checkpoint loads at any
    world size), the step counter, the dynamic ray count, the ray sampler's generator state"""
    opt = trainer.opt.state_dict()
    if trainer.sharded is not None:
        opt["step_count"] = trainer.sharded.step_count
        opt["moments"] = trainer.sharded.gather_moments(trainer.opt.tcnn_modules)
    # every rank samples its own rays (nsr.parallel.shard_seed): the sampler states are saved PER RANK (a collective, like the
    # moments above) -- restoring rank 0's state everywhere would make all ranks draw the same batch after a resume
    gen_state = trainer.gen.get_state()
    world = int(getattr(trainer, "world_size", 1))
    if world > 1 and dist.is_available() and dist.is_initialized():
        states = [None] * world
        dist.all_gather_object(states, gen_state.cpu())
    else:
        states = [gen_state]
    st = {"optimizer": opt, "global_step": int(trainer.global_step), "train_num_rays": int(trainer.train_num_rays),
          "generator": states[0], "generators": states, "extra": [o.state_dict() for o in extra_optimizers]}
    a = getattr(trainer, "_as", None)
    if a is not None:  # asynchronous mode keeps the dynamic ray count on the device
        st["train_num_rays"] = int(a["n_rays"].item())
    return st


def restore_training_state(trainer, st, extra_optimizers=()):
    trainer.opt.load_state_dict(st["optimizer"])
    if trainer.sharded is not None:
        trainer.sharded.load(trainer.opt.tcnn_modules, st["optimizer"]["moments"], st["optimizer"]["step_count"])
    for o, sd in zip(extra_optimizers, st.get("extra", [])):
        o.load_state_dict(sd)
    trainer.global_step = int(st["global_step"])
    trainer.train_num_rays = int(st["train_num_rays"])
    world, rank = int(getattr(trainer, "world_size", 1)), int(getattr(trainer, "rank", 0))
    states = st.get("generators") or ([st["generator"]] if st.get("generator") is not None else [])
    if len(states) == world and states[rank] is not None:
        trainer.gen.set_state(states[rank].cpu())  # same world size: every rank continues its own stream
    elif world > 1:
        # resumed at another world size: no saved stream belongs to this rank -- a fresh per-rank stream, distinct across ranks
        # and across resume points (never rank 0's state on every rank: all ranks would then sample identical rays)
        trainer.gen.manual_seed(shard_seed(int(getattr(trainer, "seed", 0)) + 7919 * int(st["global_step"]), rank))
    elif states:
        trainer.gen.set_state(states[0].cpu())
    a = getattr(trainer, "_as", None)
    if a is not None:
        a["n_rays"].fill_(trainer.train_num_rays)


def resync_after_model_load(trainer):
    """``model.load_state_dict()`` on a model a trainer was already built around: without this the optimizer's fp16 images
    (and, at world > 1, the sharded fp32 masters the next all-gather is cast from) would silently revert the loaded weights"""
    def hook(module, incompatible):
        trainer.opt.resync()
        if trainer.sharded is not None:
            trainer.sharded.load(trainer.opt.tcnn_modules, None, None)
    trainer.model.register_load_state_dict_post_hook(hook)


def guard_stale_state_dict(model, sharded):
    """a plain ``model.state_dict()`` in the middle of a multi-GPU run would silently hold stale table values on every
    rank: refuse it, ``Trainer.state_dict()`` gathers first"""
    def hook(module, prefix, keep_vars):
        if not sharded.master_current:
            raise RuntimeError("the fp32 hash tables of a multi-GPU run live in the ranks' optimizer shards: call "
                               "trainer.state_dict() / trainer.save() (a collective: on every rank) instead of "
                               "model.state_dict()")
    model.register_state_dict_pre_hook(hook)


def next_capacity(cap, window_max, n_rays, slots, dropped, granule=16384, floor=65536):
    """Sample-buffer capacity for the next steps of the asynchronous trainer, from LAGGED statistics (the host never
    waits for a count): 1.5 x the largest count of the last window, scaled by the room the dynamic ray count still has
    to climb (counts grow with it), rounded up to ``granule``.  Grows as soon as the window maximum passes 85 % of the
    current capacity; shrinks only when the buffers are more than twice too large and nothing was dropped."""
    if window_max <= 0:
        return cap
    room = slots / max(min(n_rays, slots), 1) if n_rays > 0 else 1.0
    want = max(-(-int(1.5 * room * window_max) // granule) * granule, floor)
    if window_max > 0.85 * cap or (not dropped and want < 0.5 * cap):
        return want
    return cap


class LazyLoss:
    """the loss of the most recent asynchronous step, formed only when somebody looks: three tiny torch kernels per
    step (mul, clamp, div) were 4 % of the step.  Reads the step's accumulator, so look before the next step runs."""

    def __init__(self, acc, trainer):
        self.acc, self.trainer, self.step = acc, trainer, trainer.global_step

    def tensor(self):
        if self.trainer.global_step != self.step:
            raise RuntimeError("the loss of an asynchronous step lives in that step's accumulator: read it (float(), "
                               ".tensor()) before the next train_step()")
        return self.acc[0] / torch.clamp(3.0 * self.acc[1], min=1.0)

    def __float__(self):
        return float(self.tensor())

    def item(self):
        return float(self)

    def isfinite(self):
        return torch.isfinite(self.tensor())


def multistep_lr_scale(step, milestones=(10000, 15000, 18000), gamma=0.33):
    """configs/nerf-blender.yaml:80-85"""
    return gamma ** sum(step >= m for m in milestones)


# The forms of the asynchronous NeRF step (csrc/step.hip nsr_nerf_step_variant keys 0, 2, 5, 9 + the host-side switches), for
# same-process A/B runs: bench.py's `step_forms_ab` block and tools/step_variants.py run windows of steps under each.
# CURRENT_FORMS: both networks' data gradients in one kernel, sample-partitioned compositing, fork events riding on kernels,
# the packing folded into the kept-row copy, the weights wait in front of the density MLP, capped weight-gradient grids.
CURRENT_FORMS = dict(keys={0: 1, 2: 1, 5: 1}, defer_pack=True, defer_weights_wait=True, wgrad_max_blocks=128)
ROUND4_FORMS = dict(keys={0: 0, 2: 0, 5: 0}, defer_pack=False, defer_weights_wait=False, wgrad_max_blocks=0)


def set_step_forms(trainer, forms):
    """switch a (fused, asynchronous) trainer's step between the forms above; synchronises (the knobs are read per launch)"""
    trainer.settle()
    torch.cuda.synchronize()
    for k, v in forms["keys"].items():
        _lib.nsr_nerf_step_variant(int(k), int(v))
    _lib.nsr_nerf_step_variant(9, int(forms["wgrad_max_blocks"]))  # (0: the pass leaves the library's cap of 512 alone)
    trainer.fused.defer_pack = bool(forms["defer_pack"])
    trainer.defer_weights_wait = bool(forms["defer_weights_wait"])


_GUARD_DEFAULT = not os.environ.get("NSR_NO_OVERFLOW_GUARD")  # (A/B switch, read once)


class Trainer:
    def __init__(self, model, dataset, config, rank=0, world_size=1, seed=42, fused=True, async_mode=False):
        self.model, self.dataset, self.config = model, dataset, config
        self.rank, self.world_size = rank, world_size
        self.device = next(model.parameters()).device
        self.train_num_samples = config["train_num_rays"] * config["num_samples_per_ray"]  # systems/nerf.py:27
        self.train_num_rays = config["train_num_rays"]
        self.global_step = 0
        self.seed = int(seed)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(shard_seed(seed, rank))  # per-rank ray batches (see module docstring)
        if world_size > 1:
            broadcast_parameters(model)
        for m in model.modules():  # grads leave the fused modules in fp32: no GradScaler, no fp16 underflow
            if isinstance(m, tcnn.Module):
                m.dtype = torch.float32
        tc = [m for m in model.modules() if isinstance(m, tcnn.Module)]
        tc_params = {id(m.params) for m in tc}
        other = [p for p in model.parameters() if id(p) not in tc_params]
        self.opt = FusedAdamW(tc, other)
        # world > 1: reduce-scatter -> AdamW on this rank's 1/P of the table -> all-gather of the fp16 image (nsr/parallel.py)
        self.sharded, self._xchg = None, None
        # (NSR_FORCE_SHARDED=1: take the multi-GPU exchange with a process group of ANY size -- a one-rank nccl group runs every
        # RCCL call of the path on a one-GPU box, tests/test_gpu_nccl_single_rank.py)
        if (world_size > 1 or os.environ.get("NSR_FORCE_SHARDED")) and not other and dist.is_initialized():
            # the table is exchanged in NSR_EXCHANGE_GROUPS (1..4) ranges cut at level boundaries (default 2: the five finest
            # levels, launched first by the fused step, travel while the other eleven are still being accumulated); the n - 1 cut
            # levels: NSR_EXCHANGE_SPLIT_LEVELS (comma list), default 11 / 13,9 / 13,10,6 for 2 / 3 / 4 groups
            splits = {}
            n_groups = max(1, min(4, int(os.environ.get("NSR_EXCHANGE_GROUPS", "2"))))
            default_cuts = {1: [], 2: [11], 3: [13, 9], 4: [13, 10, 6]}[n_groups]
            env_cuts = os.environ.get("NSR_EXCHANGE_SPLIT_LEVELS", os.environ.get("NSR_EXCHANGE_SPLIT_LEVEL", ""))
            cuts = [int(v) for v in env_cuts.split(",") if v.strip()][:n_groups - 1] if env_cuts else default_cuts
            for m in tc:
                gd = getattr(m, "grid_desc", None)
                if gd is not None and cuts:
                    lvs = sorted({min(max(c, 1), gd.n_levels - 1) for c in cuts})
                    splits[m] = [int(gd.offset[lv]) * int(gd.n_features) for lv in lvs]
            # NSR_TRANSPORT=fp32: the table gradient travels as fp32 (the A/B of the bf16 transport: tools/train_psnr.py at
            # world 2); the backward then stores a dense fp32 gradient and the exchange takes it from there
            fp32 = os.environ.get("NSR_TRANSPORT", "bf16") == "fp32"
            self.sharded = ShardedAdamW(tc, splits=splits, transport=torch.float32 if fp32 else torch.bfloat16)
            self._force_unfused_exchange = fp32
            guard_stale_state_dict(model, self.sharded)
        self.comm_timings = None
        # asynchronous single-GPU steps: AdamW on the table inside the table backward (NSR_TABLE_ADAM_SEPARATE: A/B switch)
        self.fuse_table_update = not os.environ.get("NSR_TABLE_ADAM_SEPARATE")
        # the step's stream waits for the helper stream's optimizer launch (network weights) in front of the next density MLP
        # instead of in front of the next encode (csrc/step.hip nsr_nerf_wait_before_mlp); settle() for every other reader
        self.defer_weights_wait = not os.environ.get("NSR_WEIGHTS_WAIT_EARLY")
        # (through a weak reference, handles kept: a second trainer built around the same model must not keep this one -- and its
        # GPU buffers -- alive through the hooks; close() removes them)
        import weakref
        wself = weakref.ref(self)

        def _settle_hook(_m, _inp):
            me = wself()
            if me is not None:
                me.settle()
        self._hook_handles = [mod.register_forward_pre_hook(_settle_hook) for mod in [model] + list(model.children())]
        # developer A/B switches of the asynchronous step, read once (an environment lookup per step is host time)
        self._write_inline = bool(os.environ.get("NSR_WRITE_INLINE"))
        self._exchange_unfused = bool(os.environ.get("NSR_EXCHANGE_UNFUSED")) or getattr(self, "_force_unfused_exchange", False)
        self._step_event_always = bool(os.environ.get("NSR_STEP_EVENT_ALWAYS"))
        self.last = {}
        self.fused, self._pending, self._side, self.pipeline_march, self._n_rays_dev = None, None, None, True, None
        # use_graphs: replay the queued launches of a step from a captured HIP graph.  Correct (tests/test_gpu_fused.py)
        # but measured SLOWER on ROCm 7.2 (0.759 vs 0.735 ms/step: hipGraphLaunch re-submits every node from the host),
        # so it is off by default
        self.async_mode, self._as, self.use_graphs = bool(async_mode), None, False
        self.device_occupancy_refresh = True  # asynchronous mode: csrc/occupancy.hip instead of the torch formulation
        if config["name"] != "nerf":
            # systems/neus.py has its own loss set (L1 on comp_rgb_full, eikonal, mask BCE ...): nsr.fused_neus.NeuSTrainer
            raise NotImplementedError("nsr.trainer.Trainer runs the nerf-system step; use nsr.fused_neus.NeuSTrainer")
        if fused:
            from .fused import FusedNeRFStep
            self.fused = FusedNeRFStep(model)
            self.opt.table_grad_overwritten = True
        resync_after_model_load(self)

    # ---- checkpoints -------------------------------------------------------------------------------------------------
    def state_dict(self):
        self.settle()
        return self._state_dict()

    def _state_dict(self):
        """``model.state_dict()`` (reference key set, tests/golden/state_dict_keys.json) with every fp32 parameter current.
        Multi-GPU: the tables' fp32 master values live in the owners' shards (nsr.parallel.ShardedAdamW), so this is a
        COLLECTIVE -- every rank calls it (rank 0 alone then writes the file: ``save``)."""
        return checkpoint_state_dict(self.model, self.sharded)

    def save(self, path):
        """model + training state (optimizer moments, step counts, dynamic ray count, sampler state); a COLLECTIVE at world > 1"""
        sd, ts = self.state_dict(), training_state(self)
        if self.rank == 0:
            torch.save({"state_dict": sd, "global_step": self.global_step, "training_state": ts}, path)

    def load(self, path_or_ckpt):
        """resume (reference launch.py:112-113 ``trainer.fit(..., ckpt_path=args.resume)``): weights, optimizer state, step
        counter.  Every rank calls it."""
        ck = torch.load(path_or_ckpt, map_location=self.device) if isinstance(path_or_ckpt, (str, bytes, os.PathLike)) \
            else path_or_ckpt
        self.settle()
        self.model.load_state_dict(ck["state_dict"])  # (the post-hook re-seeds fp16 images / sharded masters)
        if ck.get("training_state") is not None:
            restore_training_state(self, ck["training_state"])
        else:
            self.global_step = int(ck.get("global_step", 0))
        self._pending = None  # batches marched ahead belong to the old run
        if self._as is not None:
            self._as["marched_upto"] = self._as["packed_upto"] = self.global_step - 1
            self._as["events"].clear()
            self._as["last_step_event"] = None  # (belongs to the run that was replaced)

    def close(self):
        """withdraw what this trainer left armed in the library / on the model: the pending weights wait (its event dies with
        the trainer) and the forward pre-hooks"""
        try:
            if getattr(self, "fused", None) is not None and getattr(self, "_as", None) is not None:
                self.settle()
                _lib.nsr_nerf_wait_before_mlp(ctypes.byref(self.fused.desc), None)
        except Exception:
            pass
        for h in getattr(self, "_hook_handles", []):
            h.remove()
        self._hook_handles = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def settle(self):
        """make the current stream wait for work the asynchronous step left pending on its helper streams (the optimizer
        launch for the network weights when ``defer_weights_wait``): called before anything but the next training step reads
        the parameters -- evaluation, checkpoints, the occupancy refresh"""
        a = self._as
        if a is not None and a.get("weights_event") is not None:
            torch.cuda.current_stream().wait_event(a["weights_event"])
            _check(_lib.nsr_nerf_wait_before_mlp(ctypes.byref(self.fused.desc), None), "nsr_nerf_wait_before_mlp")
            a["weights_event"] = None

    def _all_reduce_grads(self):
        if self.world_size > 1 and self.sharded is None:
            all_reduce_gradients(list(self.model.parameters()))

    def _exchange(self):
        """(NsrTableExchange, density-MLP gradient view, colour-MLP gradient view) + the events behind each level group:
        the fused asynchronous step hands its gradients to nsr.parallel.ShardedAdamW in pieces, as they become final"""
        if self._xchg is not None:
            return self._xchg
        from nsr_hip import NsrTableExchange
        ewn, tex, sh = self.fused.ewn, self.fused.tex, self.sharded
        gd = ewn.grid_desc
        F, L = int(gd.n_features), int(gd.n_levels)
        send = sh.send_buffer(ewn)
        assert send.dtype == torch.bfloat16, "the table backward writes bf16 (ShardedAdamW transport)"
        # level run of every exchange range (highest offsets first): a range is final once the levels that reach into it are
        offs = [int(gd.offset[l]) * F for l in range(L + 1)]
        groups, hi = [], L
        for a, _ in sh.ranges(ewn)[:-1]:
            lo = max(l for l in range(L) if offs[l] <= a)  # the level a (rounded-up) cut falls into is launched with the
            # EARLIER group: its tail lies in this range, its head in the next one, which is exchanged after both groups
            groups.append((lo, hi))
            hi = lo
        groups.append((0, hi))
        groups = [g for g in groups if g[1] > g[0]]
        assert 1 <= len(groups) <= 4 and len(groups) == len(sh.ranges(ewn))
        xd = NsrTableExchange()
        xd.grad_bf16, xd.grad_bf16_elems, xd.n_groups = send.data_ptr(), send.numel(), len(groups)
        events = [torch.cuda.Event(enable_timing=True) for _ in range(len(groups) + 1)]
        for e in events:
            e.record()  # (materialises the HIP event: the C side records it from now on)
        for k, (lo, hi) in enumerate(groups):
            xd.level_begin[k], xd.level_end[k], xd.event_group[k] = lo, hi, events[k].cuda_event
        xd.event_small = events[-1].cuda_event
        self._xchg = dict(desc=xd, g_density=sh.small_grad_view(ewn), g_color=sh.small_grad_view(tex),
                          ready={"small": events[-1], ewn: events[:-1]}, groups=groups, events=events)
        return self._xchg

    def _optimizer_step(self, device_schedule, exchanged=False):
        if self.sharded is not None:  # the exchange is part of the step
            kw = {}
            if exchanged:  # the fused step already wrote bf16 table gradients / flattened MLP gradients (see _exchange)
                x = self._xchg
                kw = dict(ready=x["ready"], prefilled=(self.fused.ewn,), direct_small=(self.fused.ewn, self.fused.tex))
            elif self.fused is not None:
                kw = dict(overwritten=(self.fused.ewn,))
            self.sharded.step(lr_scale=multistep_lr_scale(self.global_step), timings=self.comm_timings, **kw)
        elif device_schedule:
            self.opt.step_device()
        else:
            self.opt.step(lr_scale=multistep_lr_scale(self.global_step))

    def loss_fn(self, out, rgb, fg):
        valid = out["rays_valid"][..., 0]
        return F.smooth_l1_loss(out["comp_rgb"][valid], rgb[valid])  # systems/nerf.py:97

    def train_step(self):
        if self.fused is not None:
            return self._train_step_async() if self.async_mode else self._train_step_fused()
        model = self.model
        with _ops.timed("phase:sample_rays"):
            rays, rgb, fg, bg = self.dataset.sample_rays(self.train_num_rays, self.gen, self.config["background_color"])
        model.background_color = bg
        with _ops.timed("phase:occupancy_update"):
            model.update_step(0, self.global_step)
            if self.world_size > 1 and self.config.get("grid_prune") and self.global_step % 16 == 0:
                sync_occupancy_grid(model.occupancy_grid)  # (every step variant keeps the ranks' grids equal)
        with _ops.timed("phase:forward"):
            out = model(rays)
        n_samples = int(out["num_samples_full" if "num_samples_full" in out else "num_samples"].sum().item())
        if self.config["dynamic_ray_sampling"] and n_samples > 0:  # systems/nerf.py:93-95
            t = int(self.train_num_rays * (self.train_num_samples / n_samples))
            self.train_num_rays = min(int(self.train_num_rays * 0.9 + t * 0.1), self.config["max_train_num_rays"])
        with _ops.timed("phase:loss"):
            loss = self.loss_fn(out, rgb, fg)
        with _ops.timed("phase:backward"):
            loss.backward()
        with _ops.timed("phase:all_reduce"):
            self._all_reduce_grads()
        with _ops.timed("phase:optimizer"):
            self._optimizer_step(False)
        self.global_step += 1
        self.last = {"loss": loss.detach(), "n_rays": rays.shape[0], "n_samples": n_samples}
        return self.last

    def _train_step_fused(self):
        """same step through nsr.fused.FusedNeRFStep (hand-chained backward, ~40 launches).

        Marching depends only on the rays and the occupancy grid, not on the parameters, and it is a latency-bound
        kernel that occupies ~3 % of the chip.  So the marching pass of step k+1 runs on a SIDE STREAM underneath step
        k's forward/backward/optimizer.  Its only input from step k is the dynamic ray count (systems/nerf.py:93-95),
        which therefore lives ON THE DEVICE: the ray arrays always have ``max_train_num_rays`` slots, slots beyond the
        count are dead rays, and a one-thread kernel behind step k's pruning pass updates the count.  The host can then
        queue step k+1's marching before it has seen step k's sample count -- the GPU starts it the moment the pruning
        pass retires.  Steps that refresh the occupancy grid (every 16th) march in order instead."""
        from .fused import FusedNeRFStep, prepare_train_rays
        model, fused, cfg = self.model, self.fused, self.config
        dynamic = bool(cfg["dynamic_ray_sampling"])
        slots = cfg["max_train_num_rays"] if dynamic else self.train_num_rays
        if self._n_rays_dev is None:
            self._n_rays_dev = torch.tensor([self.train_num_rays], dtype=torch.int32, device=self.device)
        pending, self._pending = getattr(self, "_pending", None), None
        with _ops.timed("phase:occupancy_update"):
            model.update_step(0, self.global_step)
            if self.world_size > 1 and cfg["grid_prune"] and self.global_step % 16 == 0:
                sync_occupancy_grid(model.occupancy_grid)
        if pending is None:
            with _ops.timed("phase:sample_rays"):
                rays, ro, rd, rgb, fg, bg, t_min, t_max = prepare_train_rays(
                    self.dataset, slots, self.gen, model, cfg["background_color"], n_active=self._n_rays_dev)
                handle = fused.march_begin(ro, rd, t_min, t_max)
        else:
            rays, rgb, bg, handle = pending
        model.background_color = bg
        n_live = min(self.train_num_rays, slots)  # host mirror of the device-side count for THIS batch
        next_updates_grid = cfg["grid_prune"] and (self.global_step + 1) % 16 == 0

        def before_sync(total):
            # queued behind the pruning pass, before the host waits for its count
            if dynamic and total is not None:
                with torch.cuda.device(self.device):
                    _check(_lib.nsr_update_ray_count(_ptr(total), _ptr(self._n_rays_dev), int(self.train_num_samples),
                                                     int(cfg["max_train_num_rays"]), None, _stream_ptr()),
                           "nsr_update_ray_count")
            if not self.pipeline_march or next_updates_grid:
                return
            main = torch.cuda.current_stream()
            if self._side is None:
                self._side = _shared_stream(self.device, "side")
            ev = torch.cuda.Event()
            ev.record(main)             # covers everything up to the pruning pass + count update of THIS step --
            self._side.wait_event(ev)   # not its main pass / backward / optimizer, which are queued later
            with torch.cuda.stream(self._side):
                nrays, ro, rd, nrgb, _, nbg, t_min, t_max = prepare_train_rays(
                    self.dataset, slots, self.gen, model, cfg["background_color"], n_active=self._n_rays_dev)
                h = fused.march_begin(ro, rd, t_min, t_max)
            for x in (nrays, nrgb, nbg):
                x.record_stream(main)
            self._pending = (nrays, nrgb, nbg, h)

        def after_prune(n_samples):
            if dynamic and n_samples > 0:  # systems/nerf.py:93-95 (the device runs the same arithmetic)
                t = int(self.train_num_rays * (self.train_num_samples / n_samples))
                self.train_num_rays = min(int(self.train_num_rays * 0.9 + t * 0.1), cfg["max_train_num_rays"])

        res = fused.forward_backward(rays, rgb, bg, march_handle=handle, after_prune=after_prune,
                                     before_sync=before_sync)
        n_samples = res["num_samples"]  # already on the host (the pruning sync): no extra .item()
        with _ops.timed("phase:all_reduce"):
            self._all_reduce_grads()
        with _ops.timed("phase:optimizer"):
            self._optimizer_step(False)
        self.global_step += 1
        self.last = {"loss": FusedNeRFStep.loss_value(res), "n_rays": n_live, "n_samples": n_samples}
        return self.last

    # ---- asynchronous mode: no host synchronisation inside a step ------------------------------------------------
    def _async_state(self):
        if self._as is not None:
            return self._as
        cfg, dev = self.config, self.device
        dynamic = bool(cfg["dynamic_ray_sampling"])
        slots = cfg["max_train_num_rays"] if dynamic else self.train_num_rays
        a = dict(slots=slots, pending=False, event=None, check_every=8, last_check=None)
        a["n_rays"] = torch.tensor([self.train_num_rays], dtype=torch.int32, device=dev)
        a["sets"] = [self.fused.async_ray_set(slots, dev) for _ in range(2)] if self.use_graphs else None
        a["stats"] = torch.zeros(16, dtype=torch.int32, device=dev)  # [0:6] marched, [8:14] kept (pack_from_counts_capped)
        a["rays_accum"] = torch.zeros(1, dtype=torch.int64, device=dev)
        a["host"] = torch.zeros(16, dtype=torch.int32).pin_memory()
        a["host_event"] = None
        # generous first capacities; they follow 1.5 x the largest count seen (checked every few steps, never synced)
        a["m_cap"] = max(1 << 20, 4 * self.train_num_samples)
        a["s_cap"] = max(1 << 18, (3 * self.train_num_samples) // 2)
        a["truncated"] = 0
        a["graphs"], a["eager_seen"], a["total_kept"] = {}, set(), None
        a["sets3"], a["window"] = None, 16  # ring of ray sets: one occupancy window (allocated at the first step)
        a["events"], a["marched_upto"], a["packed_upto"], a["march_stream"] = {}, -1, -1, {}
        g = self.model.occupancy_grid.binary
        a["bricks"] = torch.empty(int(_lib.nsr_grid_bricks_words64(*[int(v) for v in g.shape])), dtype=torch.int64,
                                  device=dev)
        self._as = a
        return a

    def _async_capacities(self, a):
        """follow the sample counts WITHOUT waiting for them: every ``check_every`` steps the statistics written by
        the packing kernels are copied to pinned memory behind the step; the copy of the previous round is read here."""
        if self.global_step % a["check_every"] != 0:
            return
        if a["host_event"] is not None and a["host_event"].query():
            h = a["host"]
            max_m, max_s, trunc = int(h[1]), int(h[9]), int(h[2]) + int(h[10])
            dropped = trunc != a["truncated"]
            a["m_cap"] = next_capacity(a["m_cap"], max_m, int(h[15]), a["slots"], dropped)
            a["s_cap"] = next_capacity(a["s_cap"], max_s, int(h[15]), a["slots"], dropped)
            a["truncated"] = trunc
        a["stats"][15:16].copy_(a["n_rays"])
        a["host"].copy_(a["stats"], non_blocking=True)
        a["host_event"] = torch.cuda.Event()
        a["host_event"].record(torch.cuda.current_stream())
        a["stats"][1].zero_()  # the window maxima restart AFTER the copy (stream order)
        a["stats"][9].zero_()

    def counters(self):
        """totals since the first asynchronous step (synchronises): marched / kept samples, rays, truncated launches"""
        a = self._async_state()
        self.settle()
        torch.cuda.synchronize(self.device)
        st = a["stats"].cpu()
        u64 = lambda lo, hi: (int(lo) & 0xffffffff) | ((int(hi) & 0xffffffff) << 32)
        return {"marched": u64(st[4], st[5]), "samples": u64(st[12], st[13]), "rays": int(a["rays_accum"].item()),
                "truncated": int(st[2]) + int(st[10]), "m_cap": a["m_cap"], "s_cap": a["s_cap"]}

    def _train_step_async(self):
        if self.use_graphs:
            return self._train_step_async_graphable()
        g = self._overflow_guard_state()
        if g is None:
            return self._train_step_async_ahead()
        # Lightning's precision-16 protocol (reference configs/nerf-blender.yaml:103) on the device: registered around THIS
        # trainer's launches only (the library reads it when a launch is queued)
        _lib.nsr_overflow_guard(ctypes.c_void_p(g.data_ptr()), self.fused.grad_scale)
        try:
            return self._train_step_async_ahead()
        finally:
            _lib.nsr_overflow_guard(None, 0.0)

    def _overflow_guard_state(self):
        """int32[8] on the device: {found-inf flag of even / odd steps, loss scale (float bits), clean steps, skipped steps,
        growth interval, -, -} -- torch.cuda.amp.GradScaler's state (init_scale = the step's grad_scale, backoff 0.5, growth 2
        every 2,000 clean steps) kept where the kernels read it: an overflowing data gradient skips the WHOLE optimizer step
        (table, both MLPs, moments, fp16 images) and halves the scale without the host ever looking.  One GPU, table update
        fused into the table backward (the sharded exchange does not carry the flag across ranks yet); None: off."""
        if not getattr(self, "overflow_guard", _GUARD_DEFAULT) or self.world_size > 1 or not self.fuse_table_update:
            return None
        g = getattr(self, "_guard", None)
        if g is None:
            import struct
            bits = struct.unpack("<i", struct.pack("<f", float(self.fused.grad_scale)))[0]
            g = self._guard = torch.tensor([0, 0, bits, 0, 0, int(getattr(self, "overflow_growth_interval", 2000)), 0, 0],
                                           dtype=torch.int32, device=self.device)
        return g

    def overflow_guard_stats(self):
        """(loss scale, clean steps since its last change, skipped steps) -- synchronises"""
        g = self._overflow_guard_state()
        if g is None:
            return None
        self.settle()
        v = g.cpu()
        return {"scale": float(v[2:3].view(torch.float32)[0]), "clean_steps": int(v[3]), "skipped_steps": int(v[4])}

    def _helper_stream(self):
        """the C orchestration's helper stream (csrc/step.hip) as a torch stream, or None (NSR_ADAM_ON_MAIN: A/B switch)"""
        if getattr(self, "_helper", None) is None and not os.environ.get("NSR_ADAM_ON_MAIN"):
            ptr = _lib.nsr_nerf_helper_stream()
            self._helper = torch.cuda.ExternalStream(ptr, device=self.device) if ptr else None
        return getattr(self, "_helper", None)

    def _train_step_async_ahead(self):
        """The asynchronous step with the marching passes batched over the occupancy WINDOW.

        The marching pass (~0.45 ms of latency-bound work: one dependent chain per ray) needs rays and occupancy bricks
        only, and the bricks change every 16th step.  So right after a refresh (and at the first step) the rays of the
        whole window -- up to 15 future steps, ``max_train_num_rays`` slots each -- are prepared and marched by ONE launch
        on the side stream (1,920 wavefronts instead of 15 launches of 128), while the window's first step marches its
        own set in order.  What needs the dynamic ray count of the previous step is merely which slots count; that is
        applied by the tiny packing kernel (dead slots keep nothing) queued behind the previous pruning pass.  16 ray
        sets rotate (set u % 16 was last read by step u - 16)."""
        model, fused, cfg = self.model, self.fused, self.config
        a = self._async_state()
        dynamic = bool(cfg["dynamic_ray_sampling"])
        t = self.global_step
        with _ops.timed("phase:occupancy_update"):
            if self.device_occupancy_refresh and cfg["grid_prune"] and not cfg["learned_background"]:
                for part in (model.geometry, model.texture):  # models/nerf.py:45-55, grid refresh kept on the device
                    hook = getattr(part, "update_step", None)
                    if hook is not None:
                        hook(0, t)
                _ops.grid_bricks(model.occupancy_grid.binary, out=a["bricks"])  # first packing (cached afterwards)
                if t % 16 == 0:
                    fused.refresh_occupancy_async(t, a["bricks"])
                    if self.world_size > 1:
                        sync_occupancy_grid(model.occupancy_grid)  # (DDP's broadcast_buffers: rank 0's grid everywhere)
            else:
                model.update_step(0, t)
                if self.world_size > 1 and cfg["grid_prune"] and t % 16 == 0:
                    sync_occupancy_grid(model.occupancy_grid)
        _ops.grid_bricks(model.occupancy_grid.binary, out=a["bricks"])  # re-packed in place after a grid refresh
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = _shared_stream(self.device, "side")
        side = self._side
        if a.get("bricks_event") is None or (cfg["grid_prune"] and t % 16 == 0):
            a["bricks_event"] = torch.cuda.Event()
            a["bricks_event"].record(main)  # marching passes queued from now on read the re-packed bricks
        W = a["window"]
        if a["sets3"] is None:  # first asynchronous step (or use_graphs was switched off)
            a["sets3"] = self.fused.async_ray_sets(W, a["slots"], self.device)
            a["marched_upto"] = a["packed_upto"] = t - 1
        sets, ev = a["sets3"], a["events"]
        stats_m, stats_s = a["stats"][0:8], a["stats"][8:16]
        ring = a.get("event_ring")
        if ring is None:  # (torch events are re-used: an entry of ``ev`` lives for at most 3 steps, creating one costs microseconds)
            ring = a["event_ring"] = [[torch.cuda.Event() for _ in range(16)], 0]

        def new_event():
            ring[1] = (ring[1] + 1) & 15
            return ring[0][ring[1]]
        refresh = lambda u: bool(cfg["grid_prune"]) and u % 16 == 0  # step u marches through a grid refreshed at its start

        def queue_march(u0, u1, stream):
            """steps u0 .. u1-1 (consecutive ring slots) in one launch"""
            with torch.cuda.stream(stream):
                done = a.get("last_step_event")
                if done is not None and stream is not main:
                    stream.wait_event(done)  # ring slots u % W were last read by steps u - W <= the last queued step
                stream.wait_event(a["bricks_event"])
                fused.march_async_many([sets[u % W] for u in range(u0, u1)], self.dataset, self.gen,
                                       cfg["background_color"], bricks=a["bricks"])
                e = torch.cuda.Event()
                e.record(stream)
            for u in range(u0, u1):
                ev[("march", u)] = e
            a["marched_upto"] = u1 - 1

        def queue_pack(u, stream):
            # (launches take the raw stream pointer, events their stream explicitly: a ``with torch.cuda.stream(...)`` block
            # costs the host ~25 us per step in device-index / current-stream bookkeeping)
            sp = ctypes.c_void_p(stream.cuda_stream)
            pruned = ev.get(("prune", u - 1))
            if pruned is not None:
                stream.wait_event(pruned)  # the ray count of step u is final behind step u - 1's pruning pass
            marched = ev.get(("march", u))
            if marched is not None:
                stream.wait_event(marched)
            fused.pack_async(sets[u % W], a["n_rays"], a["m_cap"], stats_m, stream=sp)
            if stream is not main and not self._write_inline:  # ... and the sample arrays + positions, off the step's own chain
                fused.write_async(sets[u % W], consumer_stream=main, stream=sp, writer_stream=stream)
            e = new_event()
            e.record(stream)
            ev[("pack", u)] = e
            a["packed_upto"] = u

        def window_end(u):  # first step after u that marches through a NEW grid (exclusive end of u's window)
            return (u // 16 + 1) * 16 if cfg["grid_prune"] else u + W

        if a["marched_upto"] < t:     # first step, or the step right after a grid refresh: its own set in order ...
            with _ops.timed("phase:sample_rays"):
                queue_march(t, t + 1, main)
            if self.pipeline_march:   # ... and the rest of the window in ONE launch on the side stream
                u1 = min(window_end(t), t + W)
                u0 = t + 1
                while u0 < u1:        # a run of ring slots must not wrap
                    stop = min(u1, u0 + (W - u0 % W))
                    queue_march(u0, stop, side)
                    u0 = stop
        if a["packed_upto"] < t:
            queue_pack(t, main)
        main.wait_event(ev[("pack", t)])
        rs = sets[t % W]
        model.background_color = rs["bg"]

        def after_prune_queued(total, e):
            # (called once the whole step is queued; ``e`` was recorded on the main stream right behind the pruning pass)
            # the ray-count update feeds only the NEXT step's packing: it runs on the side stream, off the main queue
            stream = side if self.pipeline_march else main
            sp = ctypes.c_void_p(stream.cuda_stream)
            if e is not None:
                stream.wait_event(e)
            elif stream is not main:  # the main pass's own event behind its kept-row copy (csrc/step.hip)
                _check(_lib.nsr_nerf_wait_kept_rows(sp), "nsr_nerf_wait_kept_rows")
            _check(_lib.nsr_update_ray_count(_ptr(total), _ptr(a["n_rays"]),
                                             int(self.train_num_samples) if dynamic else 0,
                                             int(cfg["max_train_num_rays"]), _ptr(a["rays_accum"]), sp),
                   "nsr_update_ray_count")
            e2 = new_event()
            e2.record(stream)
            ev[("prune", t)] = e2  # "the ray count of step t + 1 is final"
            if self.pipeline_march and a["marched_upto"] >= t + 1 and a["packed_upto"] < t + 1:
                queue_pack(t + 1, side)           # the only work between this pruning pass and the next one

        # one GPU: AdamW on the hash table happens inside the table backward (no gradient store / optimizer read-back)
        fuse_table = self.world_size == 1 and self.sharded is None and self.fuse_table_update
        xchg = self._exchange() if (self.sharded is not None and not self._exchange_unfused) else None
        res = fused.forward_backward_async(rs, a["s_cap"], stats_s, after_prune_queued=after_prune_queued,
                                           table_adam=self.opt.table_update_desc(fused.ewn) if fuse_table else None,
                                           exchange=(xchg["desc"], xchg["g_density"], xchg["g_color"]) if xchg else None,
                                           # (this trainer joins the helper stream itself, behind its optimizer launch)
                                           defer_wgrad_join=bool(fuse_table and self._helper_stream() is not None
                                                                 and not os.environ.get("NSR_WGRAD_INLINE")))
        a["total_kept"] = res["num_samples"]
        with _ops.timed("phase:all_reduce"):
            self._all_reduce_grads()
        with _ops.timed("phase:optimizer"):
            if fuse_table and self._helper_stream() is not None:
                # what is left for the optimizer (the MLP weights: one launch that also advances the device-side schedule) runs
                # on the main pass's helper stream, right behind the weight-gradient kernels it reads from -- underneath the
                # table backward, off the step's own chain; the main stream only waits for its event
                hs = self._helper
                self.opt.step_device(skip_table_of=fused.ewn, other_stream_reads=True, stream=ctypes.c_void_p(hs.cuda_stream))
                ev_opt = a.setdefault("opt_events", [torch.cuda.Event() for _ in range(4)])[t % 4]
                ev_opt.record(hs)
                if self.defer_weights_wait and not (cfg["grid_prune"] and (t + 1) % 16 == 0):
                    # the next step's stream waits for the new network weights between its encode and its density MLP, not in
                    # front of the encode (csrc/step.hip nsr_nerf_wait_before_mlp); anything else that reads them: settle()
                    _check(_lib.nsr_nerf_wait_before_mlp(ctypes.byref(fused.desc), ctypes.c_void_p(ev_opt.cuda_event)),
                           "nsr_nerf_wait_before_mlp")
                    a["weights_event"] = ev_opt
                else:  # (the next step starts with the occupancy refresh, which evaluates the density network)
                    main.wait_event(ev_opt)
                    a["weights_event"] = None
            elif fuse_table:
                self.opt.step_device(skip_table_of=fused.ewn)
            else:
                self._optimizer_step(True, exchanged=xchg is not None)
        if a["marched_upto"] < t + 1 or self._step_event_always:
            # (only a step whose successor queues a marching launch on the side stream needs the marker -- an event record on
            # the main stream costs ~9 us of the step's chain)
            a["last_step_event"] = torch.cuda.Event()
            a["last_step_event"].record(main)
        for key in [k for k in ev if k[1] < t - 2]:
            del ev[key]
        self.global_step += 1
        self._async_capacities(a)
        self.last = {"loss": LazyLoss(res["loss_acc"], self), "n_rays": a["n_rays"], "n_samples": a["total_kept"]}
        return self.last

    def _train_step_async_graphable(self):
        """the fused step with every count on the device (FusedNeRFStep.forward_backward_async): the host only queues
        work and never waits for the GPU.  With ``use_graphs`` the queued launches of a step are captured once per
        variant (ray-set parity x in-order marching x launches-next-marching x capacities) as a HIP graph and
        replayed: ~45 kernel launches become one graph launch.  Returns device tensors; ``counters()`` gives totals."""
        model, cfg = self.model, self.config
        a = self._async_state()
        if a["sets"] is None:  # use_graphs was switched on after the first step
            a["sets"] = [self.fused.async_ray_set(a["slots"], self.device) for _ in range(2)]
            a["pending"] = False
        with _ops.timed("phase:occupancy_update"):
            model.update_step(0, self.global_step)
            if self.world_size > 1 and cfg["grid_prune"] and self.global_step % 16 == 0:
                sync_occupancy_grid(model.occupancy_grid)
        _ops.grid_bricks(model.occupancy_grid.binary, out=a["bricks"])  # re-packed in place after a grid refresh
        k = self.global_step & 1
        inorder = not a["pending"]
        launch_next = bool(self.pipeline_march and not (cfg["grid_prune"] and (self.global_step + 1) % 16 == 0))
        key = (k, inorder, launch_next, a["m_cap"], a["s_cap"])
        graphs_ok = self.use_graphs and self.world_size == 1 and not _ops.profiling()  # events cannot be captured
        graph = a["graphs"].get(key) if graphs_ok else None
        if graph is None and graphs_ok and key in a["eager_seen"]:
            graph = self._capture_async(a, key)
        if graph is not None:
            graph[0].replay()
            a["pending"] = launch_next
            loss = graph[1]
        else:
            loss = self._async_body(a, k, inorder, launch_next)
            a["eager_seen"].add(key)
        self.global_step += 1
        self._async_capacities(a)
        self.last = {"loss": LazyLoss(loss, self), "n_rays": a["n_rays"], "n_samples": a["total_kept"]}
        return self.last

    def _capture_async(self, a, key):
        k, inorder, launch_next = key[:3]
        if len(a["graphs"]) > 16:  # capacities moved a lot: drop the stale captures
            a["graphs"].clear()
        g = torch.cuda.CUDAGraph()
        if hasattr(g, "register_generator_state"):
            g.register_generator_state(self.gen)
        was_pending = a["pending"]
        with torch.cuda.graph(g):
            loss = self._async_body(a, k, inorder, launch_next)
        a["pending"] = was_pending  # capturing queued nothing
        a["graphs"][key] = (g, loss)
        return a["graphs"][key]

    def _async_body(self, a, k, inorder, launch_next):
        """everything of one step that is queued on the GPU (and nothing else): capturable"""
        from .fused import FusedNeRFStep
        model, fused, cfg = self.model, self.fused, self.config
        dynamic = bool(cfg["dynamic_ray_sampling"])
        main = torch.cuda.current_stream()
        rs = a["sets"][k]
        stats_m, stats_s = a["stats"][0:8], a["stats"][8:16]
        if inorder:
            with _ops.timed("phase:sample_rays"):
                fused.march_async(rs, self.dataset, self.gen, a["n_rays"], a["m_cap"], stats_m, cfg["background_color"],
                                  bricks=a["bricks"])
        model.background_color = rs["bg"]

        def after_prune_queued(total, _pruned=None):
            with torch.cuda.device(self.device):
                _check(_lib.nsr_update_ray_count(_ptr(total), _ptr(a["n_rays"]),
                                                 int(self.train_num_samples) if dynamic else 0,
                                                 int(cfg["max_train_num_rays"]), _ptr(a["rays_accum"]), _stream_ptr()),
                       "nsr_update_ray_count")
            a["pending"] = False
            if not launch_next:
                return
            if self._side is None:
                self._side = _shared_stream(self.device, "side")
            # (always an event recorded HERE, behind the ray-count update queued above on the main stream -- an event
            # recorded right behind the pruning pass would let the side stream's marching read the count before it is updated)
            _pruned = torch.cuda.Event()
            _pruned.record(main)
            self._side.wait_event(_pruned)
            with torch.cuda.stream(self._side):
                fused.march_async(a["sets"][1 - k], self.dataset, self.gen, a["n_rays"], a["m_cap"], stats_m,
                                  cfg["background_color"], bricks=a["bricks"])
                a["event"] = torch.cuda.Event()
                a["event"].record(self._side)
            a["pending"] = True

        # one GPU: AdamW on the hash table happens inside the table backward (no gradient store / optimizer read-back)
        fuse_table = self.world_size == 1 and self.sharded is None and self.fuse_table_update
        res = fused.forward_backward_async(rs, a["s_cap"], stats_s, after_prune_queued=after_prune_queued,
                                           table_adam=self.opt.table_update_desc(fused.ewn) if fuse_table else None)
        a["total_kept"] = res["num_samples"]
        with _ops.timed("phase:all_reduce"):
            self._all_reduce_grads()
        with _ops.timed("phase:optimizer"):
            if fuse_table:
                self.opt.step_device(skip_table_of=fused.ewn)
            else:
                self._optimizer_step(True)
        if a["pending"]:
            main.wait_event(a["event"])  # join: the next step (or the grid refresh before it) starts behind the marching
        return res["loss_acc"]  # [sum, valid rays]: the loss value itself is formed on demand (LazyLoss)
