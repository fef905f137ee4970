"""Drop-in ``models`` registry entries: the fused MI355X step BEHIND the reference's model interface.

The reference's Lightning systems never look inside a model -- ``systems/nerf.py:30-31,87-122`` does::

    out  = self.model(batch['rays'])                       # dict: comp_rgb, opacity, depth, rays_valid, num_samples, ...
    loss = F.smooth_l1_loss(out['comp_rgb'][valid], rgb[valid]) (+ distortion loss on weights / points / intervals ...)
    loss.backward(); optimizer.step()                      # by Lightning
    self.model.update_step(epoch, global_step)             # systems/base.py:54-57, every batch start
    self.model.regularizations(out) / .export(cfg) / .background_color = ...

``FusedNeRFModel`` is an ``nn.Module`` with exactly that surface and the reference's state-dict keys
(``geometry.encoding_with_network.params`` ...), whose ``forward`` is ONE ``torch.autograd.Function``: march + sigma pass +
encode + MLPs + composite as three C calls (csrc/step.hip), and whose backward takes whatever gradients the system's loss
sends to ``comp_rgb`` / ``opacity`` / ``depth`` / ``weights`` and runs the hand-chained backward (composite -> colour MLP ->
density MLP -> owner-computes table backward) as one C call.  Loss, optimizer, GradScaler, schedulers stay the caller's.
``FusedNeuSModel`` does the same for ``models.make('neus', cfg)`` (NeuS, NeuS + NeRF++ background, neuralangelo) with the
differentiable outputs the reference's NeuS system puts losses on (systems/neus.py:96-139).
A maintainer switches the reference over with one line::

    import nsr.models; nsr.models.register(models)          # models.make('nerf' | 'neus', cfg) now build the fused entries

The modular path (the reference's own models/*.py on the drop-in tinycudann / nerfacc packages) stays available; this
entry is the fast one.
"""
import os

import torch

from .state import HotPathState


def _plain(cfg):
    """OmegaConf / dict-like config -> plain nested dict (the reference hands ``config.model`` as a DictConfig)"""
    try:
        from omegaconf import OmegaConf  # noqa: WPS433  (absent in this image; present where the reference runs)
        if OmegaConf.is_config(cfg):
            return OmegaConf.to_container(cfg, resolve=True)
    except ImportError:
        pass
    if hasattr(cfg, "items"):
        return {k: _plain(v) for k, v in cfg.items()}
    if isinstance(cfg, (list, tuple)):
        return [_plain(v) for v in cfg]
    return cfg


def _fp16_backward_scale(model, under_autocast):
    """scale of dL/dy in front of the fp16 rounding of the fused MLPs' backward.  tcnn multiplies by 128 ON TOP of what arrives
    (its loss_scale for fp16 networks); under Lightning's precision-16 protocol what arrives already carries GradScaler's
    65536.  So: 128 when the forward ran under autocast (that protocol: the scaler supplies the rest; 65536 here on top of it
    overflows fp16), 65536 otherwise (no scaler in front: 128 alone leaves late-training gradients at the fp16 subnormal
    edge, nsr/fused.py).  ``model.fp16_grad_scale`` (a float) overrides the choice."""
    fixed = getattr(model, "fp16_grad_scale", None)
    if fixed is not None:
        return float(fixed)
    return 128.0 if under_autocast else 65536.0


# ---- outputs that do not stall the host (round 5) -----------------------------------------------------------------------------
# The reference's training_step (systems/nerf.py:87-99) synchronises three times on the model's outputs: ``.sum().item()`` on
# num_samples and two boolean-mask indexings with rays_valid (a nonzero each).  With ``model.lazy_outputs = True`` (OPT-IN since
# round 6 -- the default is the reference-exact synchronising entry: ``num_samples`` of THIS forward; NSR_BOUNDARY_LAZY=1 too)
# the fused forward queues its launches and returns at once; the system's OWN statements then run unchanged on:
#   * ``num_samples``: a count whose ``.sum().item()`` gives the kept samples of the PREVIOUS forward (already in pinned memory:
#     the dynamic ray count is a 0.9 / 0.1 moving average, one step of lag moves it by nothing measurable -- PSNR checked,
#     profiles/r05_psnr_boundary_lazy.json) -- ``.current()`` gives this forward's, waiting for it;
#   * ``rays_valid``: a bool tensor subclass; ``x[rays_valid[..., 0]]`` is DEFERRED (rows + mask), and F.smooth_l1_loss /
#     mse_loss / l1_loss of two such selections over the same mask are computed as masked means on the device (_MaskedLoss: two
#     launches) -- the same number as the loss of the gathered rows, no nonzero().  Anything else done to a deferred
#     selection materialises it (one synchronisation, the reference's behaviour);
#   * per-sample outputs (weights, points, intervals, ray_indices): sliced to the live count only when somebody reads them.
class _LazyCount:
    def __init__(self, handle):
        self._h = handle

    def current(self):
        return self._h.current()[1]

    def _value(self):
        prev = self._h.previous()
        return prev[1] if prev is not None else self._h.current()[1]  # (the first forward has no predecessor: its own count)

    def sum(self, *a, **k):
        return self

    def item(self):
        return int(self._value())

    __int__ = item

    def __float__(self):
        return float(self._value())

    def tensor(self):
        return torch.as_tensor([self._value()], dtype=torch.int32)

    def __getattr__(self, name):  # anything else: the one-element CPU tensor the synchronising entry returns
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.tensor(), name)

    def __index__(self):
        return int(self._value())

    def __bool__(self):
        return bool(self._value())

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        args = tuple(a.tensor() if isinstance(a, _LazyCount) else a for a in args)
        return func(*args, **(kwargs or {}))


class _ValidMask(torch.Tensor):
    """``rays_valid`` of a lazy forward: indexing another tensor's rows with it is deferred (``_MaskedRows``)"""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func is torch.Tensor.__getitem__ and len(args) == 2 and isinstance(args[1], _ValidMask) \
                and not isinstance(args[0], _ValidMask) and args[1].dim() == 1 and args[0].dim() >= 1 \
                and args[0].shape[0] == args[1].shape[0]:
            return _MaskedRows(args[0], args[1].as_subclass(torch.Tensor))
        with torch._C.DisableTorchFunctionSubclass():
            out = func(*args, **kwargs)
        if func is torch.Tensor.__getitem__ and isinstance(out, torch.Tensor) and out.dtype == torch.bool:
            return out.as_subclass(_ValidMask)  # (rays_valid[..., 0] stays a validity mask)
        return out


class _MaskedRows:
    """``base[mask]`` not yet gathered"""
    _LOSSES = None

    def __init__(self, base, mask):
        self.base, self.mask = base, mask

    def materialize(self):
        return self.base[self.mask]

    def __getattr__(self, name):  # anything the deferred form does not know: the gathered tensor's
        return getattr(self.materialize(), name)

    # (Python looks operators up on the type, not through __getattr__: arithmetic / indexing / comparisons on a deferred
    # selection gather it first, like every other use the deferred form does not know)
    def __len__(self):
        return len(self.materialize())

    def __iter__(self):
        return iter(self.materialize())

    def __getitem__(self, idx):
        return self.materialize()[idx]

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        import torch.nn.functional as F
        kwargs = dict(kwargs or {})
        if func in (F.smooth_l1_loss, F.mse_loss, F.l1_loss, F.huber_loss) and len(args) >= 2 \
                and isinstance(args[0], _MaskedRows) and isinstance(args[1], _MaskedRows) \
                and args[0].mask.data_ptr() == args[1].mask.data_ptr() and args[0].mask.shape == args[1].mask.shape \
                and kwargs.get("reduction", "mean") == "mean" and len(args) == 2 \
                and kwargs.get("size_average") is None and kwargs.get("reduce") is None:
            a, b = args
            fused = _MaskedLoss.maybe(func, a, b, kwargs)
            if fused is not None:
                return fused
            kwargs["reduction"] = "none"
            # (rows outside the mask never reach the reference's loss -- pred[valid] gathers them away: zeroed on BOTH inputs
            # before the function, so an inf / NaN there touches neither the value nor the gradients; no valid row: NaN, as
            # the mean of an empty selection is)
            mb = a.mask.view(-1, *([1] * (a.base.dim() - 1))).bool()
            zero = a.base.new_zeros(())
            per = func(torch.where(mb, a.base, zero), torch.where(mb, b.base, zero.to(b.base.dtype)), **kwargs)
            per = torch.where(mb, per, per.new_zeros(()))
            n = mb.sum() * (per.numel() // max(per.shape[0], 1))
            return per.sum() / n
        args = tuple(x.materialize() if isinstance(x, _MaskedRows) else x for x in args)
        kwargs = {k: (v.materialize() if isinstance(v, _MaskedRows) else v) for k, v in kwargs.items()}
        return func(*args, **kwargs)


def _gathering_operator(name):
    def op(self, *args):
        args = tuple(a.materialize() if isinstance(a, _MaskedRows) else a for a in args)
        return getattr(self.materialize(), name)(*args)
    op.__name__ = name
    return op


for _name in ("add radd sub rsub mul rmul truediv rtruediv floordiv rfloordiv mod rmod pow rpow matmul rmatmul neg pos abs invert "
              "and rand or ror xor rxor lt le gt ge eq ne").split():
    setattr(_MaskedRows, f"__{_name}__", _gathering_operator(f"__{_name}__"))
_MaskedRows.__hash__ = object.__hash__  # (__eq__ is elementwise, as a tensor's)


def _count_operator(name):
    def op(self, *args):
        args = tuple(a.tensor() if isinstance(a, _LazyCount) else a for a in args)
        return getattr(self.tensor(), name)(*args)
    op.__name__ = name
    return op


for _name in "add radd sub rsub mul rmul truediv rtruediv floordiv rfloordiv lt le gt ge eq ne neg".split():
    setattr(_LazyCount, f"__{_name}__", _count_operator(f"__{_name}__"))
_LazyCount.__hash__ = object.__hash__
del _name


class _MaskedLoss(torch.autograd.Function):
    """``F.smooth_l1_loss / mse_loss / l1_loss / huber_loss(pred[valid], target[valid])`` (reduction "mean") of two deferred
    selections over the same mask as two launches (``nsr_masked_loss_forward / _backward``; systems/nerf.py:97,
    systems/neus.py:98,102) instead of torch's eight elementwise / reduction kernels forward and as many backward; summed in a
    fixed order.  No valid row -> 0 (torch: NaN)."""

    @staticmethod
    def maybe(func, a, b, kwargs):
        import torch.nn.functional as F
        kinds = {F.smooth_l1_loss: (0, "beta"), F.mse_loss: (1, None), F.l1_loss: (2, None), F.huber_loss: (3, "delta")}
        pred, target, mask = a.base, b.base, a.mask
        if (func not in kinds or not pred.is_cuda or pred.dtype != torch.float32 or target.dtype != torch.float32
                or pred.shape != target.shape or pred.dim() not in (1, 2) or target.requires_grad or mask.dtype != torch.bool
                or not (pred.is_contiguous() and target.is_contiguous() and mask.is_contiguous())
                or any(v is not None for k, v in kwargs.items() if k not in ("reduction", "beta", "delta"))
                or os.environ.get("NSR_MASKED_LOSS_TORCH")):
            return None
        kind, knob = kinds[func]
        beta = float(kwargs.get(knob, 1.0)) if knob else 0.0
        if kind == 0 and beta == 0.0:
            kind = 2  # (torch: smooth-L1 with beta = 0 is L1)
        return _MaskedLoss.apply(pred, target, mask, kind, beta)

    @staticmethod
    def forward(ctx, pred, target, mask, kind, beta):
        from nsr_hip import lib, check, ptr, stream_ptr
        out = torch.empty(int(lib.nsr_masked_loss_out_floats()), dtype=torch.float32, device=pred.device)
        n, ch = int(pred.shape[0]), (int(pred.shape[1]) if pred.dim() == 2 else 1)
        with torch.cuda.device(pred.device):
            check(lib.nsr_masked_loss_forward(ptr(pred), ptr(target), ptr(mask), n, ch, kind, beta, ptr(out), stream_ptr()),
                  "nsr_masked_loss_forward")
        ctx.save_for_backward(pred, target, mask, out)
        ctx.args = (n, ch, kind, beta)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        from nsr_hip import lib, check, ptr, stream_ptr
        pred, target, mask, out = ctx.saved_tensors
        n, ch, kind, beta = ctx.args
        d = torch.empty_like(pred)
        g = g.to(torch.float32).contiguous()
        with torch.cuda.device(pred.device):
            check(lib.nsr_masked_loss_backward(ptr(pred), ptr(target), ptr(mask), n, ch, kind, beta, ptr(out), ptr(g), ptr(d),
                                               stream_ptr()), "nsr_masked_loss_backward")
        return d, None, None, None, None


class _LazyOutputs(dict):
    """the model's output dict whose per-sample entries are capacity-sized until read (then: one wait for the count).  A dict
    subclass (the reference's systems test and update the model's output as a dict), so C fast paths that bypass
    ``__getitem__`` -- ``{**out}``, ``dict(out)``, ``out.copy()``, ``out.pop(k)``, ``out.setdefault`` -- would see the
    placeholders: every one of them resolves the pending entries first."""

    def __init__(self, eager, lazy, count):
        super().__init__(eager)
        self._lazy, self._count = dict(lazy), count
        for k in self._lazy:
            dict.__setitem__(self, k, None)

    def _resolve(self, k):
        if k in self._lazy:
            S = self._count.current()
            v = self._lazy.pop(k)
            dict.__setitem__(self, k, (v() if callable(v) else v)[:S])  # (callable: derived arrays nobody may ever read)

    def _resolve_all(self):
        for k in list(self._lazy):
            self._resolve(k)

    def __getitem__(self, k):
        self._resolve(k)
        return dict.__getitem__(self, k)

    def __setitem__(self, k, v):
        self._lazy.pop(k, None)
        dict.__setitem__(self, k, v)

    def get(self, k, default=None):
        if k in self:
            return self[k]
        return default

    def pop(self, k, *default):
        self._resolve(k)
        return dict.pop(self, k, *default)

    def setdefault(self, k, default=None):
        self._resolve(k)
        return dict.setdefault(self, k, default)

    def items(self):
        self._resolve_all()
        return dict.items(self)

    def values(self):
        self._resolve_all()
        return dict.values(self)

    def keys(self):  # (``{**out}`` / ``dict(out)`` of a dict SUBCLASS that overrides keys() go through keys() + __getitem__)
        return dict.keys(self)

    def copy(self):
        self._resolve_all()
        return dict(dict.items(self))

    def __iter__(self):
        return dict.__iter__(self)

    def __or__(self, other):
        return self.copy() | other

    def __ror__(self, other):
        return other | self.copy()

    def __reduce__(self):
        return (dict, (self.copy(),))


class _RenderNeRF(torch.autograd.Function):
    """comp_rgb, opacity, depth, weights = render(rays, background; geometry params, texture params)"""

    @staticmethod
    def forward(ctx, model, need_grad, rays, background, p_geometry, p_texture, *p_empty):
        # (p_empty: the zero-element parameters of parameter-free modules -- the SH encoding -- ride along so that they receive
        # their (empty) gradient like every other parameter: DDP(find_unused_parameters=False) waits for one from each)
        ctx.empty_shapes = [tuple(p.shape) for p in p_empty]
        # (need_grad is decided by the caller: grad mode is always off inside Function.forward)
        step = model._runner()
        out, state = step.render_forward(rays.detach(), background.detach(), prepare_backward=need_grad,
                                         lazy=bool(model.training and getattr(model, "lazy_outputs", False)))
        ctx.prepared = bool(need_grad)
        ctx.step, ctx.state = step, state
        ctx.grad_scale = _fp16_backward_scale(model, torch.is_autocast_enabled())
        model._last = out  # the non-differentiable outputs (ray_indices, t_starts, t_ends, counts) for forward_()
        ctx.mark_non_differentiable(out["ray_indices"])
        return out["comp_rgb"], out["opacity"], out["depth"], out["weights"], out["ray_indices"]

    @staticmethod
    def backward(ctx, g_comp, g_opacity, g_depth, g_weights, _g_ri):
        if not ctx.prepared:
            raise RuntimeError("FusedNeRFModel: backward through a forward that ran without gradients enabled")
        if g_comp is None:  # the loss did not touch the colours: their upstream gradient is zero
            g_comp = torch.zeros((ctx.state["n_rays"], 3), device=ctx.state["ws"].device)
        ctx.step.desc.grad_scale = ctx.grad_scale
        g1, g2 = ctx.step.render_backward(ctx.state, g_comp, g_opacity, g_depth, g_weights)
        ctx.state = None  # the workspaces go back to the allocator
        return (None, None, None, None, g1, g2) + tuple(g1.new_zeros(sh) for sh in ctx.empty_shapes)


class FusedNeRFModel(HotPathState):
    """``models.make('nerf', config)`` (reference models/nerf.py:14-161) on the fused step.  Same constructor argument, same
    attributes the systems touch (``background_color``, ``randomized``, ``occupancy_grid``, ``geometry``, ``texture``,
    ``scene_aabb``, ``render_step_size``), same state-dict keys, same output dict -- except that ``num_samples`` is a CPU
    int32 tensor (the count is already on the host; the system's ``.sum().item()`` then costs no second synchronisation)."""

    def __init__(self, config):
        cfg = _plain(config)
        if cfg.get("name") != "nerf":
            raise ValueError("FusedNeRFModel builds the 'nerf' model section")
        super().__init__(cfg)
        self._step, self._last, self._bricks = None, None, None
        import os
        # False (default): the reference's behaviour -- one synchronisation per forward, num_samples a CPU tensor holding THIS
        # forward's count, plain bool rays_valid.  True (opt-in; NSR_BOUNDARY_LAZY=1): training forwards return without a host
        # synchronisation (see _LazyCount / _ValidMask above): num_samples is then one forward late
        self.lazy_outputs = bool(os.environ.get("NSR_BOUNDARY_LAZY"))

    def _runner(self):
        if self._step is None:
            from .fused import FusedNeRFStep
            self._step = FusedNeRFStep(self)
        return self._step

    # -- systems/base.py:54-57 -> models/nerf.py:45-55 ------------------------------------------------------------------
    def update_step(self, epoch, global_step):
        """occupancy refresh every 16th step (nerfacc ``every_n_step``) -- on the device, no ``torch.nonzero``"""
        cfg = self.config
        if not (self.training and cfg["grid_prune"]):
            return
        if global_step % 16 == 0:
            from nsr_hip import lib, ops
            step = self._runner()
            g = self.occupancy_grid.binary
            if self._bricks is None:
                self._bricks = torch.empty(int(lib.nsr_grid_bricks_words64(*[int(v) for v in g.shape])), dtype=torch.int64,
                                           device=g.device)
            ops.grid_bricks(g, out=self._bricks)
            step.refresh_occupancy_async(int(global_step), self._bricks)

    # -- models/nerf.py:61-127 ------------------------------------------------------------------------------------------
    def forward_(self, rays):
        bg = self.background_color if self.background_color is not None else torch.ones(3, device=rays.device)
        ewn, tex = self.geometry.encoding_with_network, self.texture.network
        need_grad = torch.is_grad_enabled() and (ewn.params.requires_grad or tex.params.requires_grad)
        empty = [p for p in self.parameters() if p.numel() == 0 and p.requires_grad]
        comp_rgb, opacity, depth, weights, ray_indices = _RenderNeRF.apply(self, need_grad, rays, bg, ewn.params, tex.params,
                                                                           *empty)
        last = self._last
        if last.get("count") is not None:  # a lazy forward: nothing here waits for the GPU
            count = _LazyCount(last["count"])
            t0, t1 = last["t_starts"], last["t_ends"]
            return _LazyOutputs({"comp_rgb": comp_rgb, "opacity": opacity, "depth": depth,
                                 "rays_valid": (opacity > 0).as_subclass(_ValidMask), "num_samples": count},
                                {"weights": weights.view(-1), "points": lambda: ((t0 + t1) / 2.0).view(-1),
                                 "intervals": lambda: (t1 - t0).view(-1), "ray_indices": ray_indices.view(-1)}, count)
        out = {"comp_rgb": comp_rgb, "opacity": opacity, "depth": depth, "rays_valid": opacity > 0,
               "num_samples": torch.as_tensor([last["num_samples"]], dtype=torch.int32)}
        if self.training:
            t0, t1 = last["t_starts"], last["t_ends"]
            out.update({"weights": weights.view(-1), "points": ((t0 + t1) / 2.0).view(-1), "intervals": (t1 - t0).view(-1),
                        "ray_indices": ray_indices.view(-1)})
        return out

    def forward(self, rays):
        if self.training:
            out = self.forward_(rays)
            return out if isinstance(out, _LazyOutputs) else {**out}
        from .export import chunk_batch
        with torch.no_grad():
            return {**chunk_batch(self.forward_, int(self.config["ray_chunk"]), True, rays)}

    def eval(self):
        self.randomized = False
        return super().eval()

    def regularizations(self, out):
        return {}  # models/geometry.py / models/texture.py: both regularizations() are empty for this model

    def isosurface(self):
        raise NotImplementedError("marching cubes is a CPU library call in the reference (out of the hot path's scope): "
                                  "nsr.export.isosurface_levels evaluates the level lattice on the device")

    @torch.no_grad()
    def export(self, export_config):
        """models/nerf.py:151-161 without the CPU mesh extraction: the level lattice (+ per-vertex colours on request)"""
        from . import export as ex
        ec = _plain(export_config) if export_config is not None else {}
        res = int(self.config["geometry"].get("isosurface", {}).get("resolution", 256)) if isinstance(
            self.config["geometry"].get("isosurface"), dict) else 256
        return {"level": ex.isosurface_levels(self, res), "export_vertex_color": bool(ec.get("export_vertex_color", False))}


class _RenderNeuS(torch.autograd.Function):
    """the differentiable outputs of ``NeuSModel.forward_`` (models/neus.py:205-287) from rays + every parameter of the model;
    backward hands the gradients the system's loss sends to them (systems/neus.py:96-139) to the fused backward"""

    KEYS = ("comp_rgb_full", "comp_rgb", "opacity", "depth", "weights", "sdf_samples", "sdf_grad_samples", "sdf_laplace_samples")

    @staticmethod
    def forward(ctx, model, need_grad, rays, background, *params):
        step = model._runner()
        res, finish = step.render(rays.detach(), background.detach(), bool(need_grad))
        model._last = res
        ctx.finish, ctx.params, ctx.step = finish, params, step
        ctx.grad_scale = _fp16_backward_scale(model, torch.is_autocast_enabled())
        ctx.set_materialize_grads(False)  # outputs the loss never touched arrive as None, not as zero tensors
        outs = [res[k] if k in res else rays.new_zeros(0) for k in _RenderNeuS.KEYS]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        params = ctx.params
        if ctx.finish is None:  # nothing was marched (or the forward ran without gradients): no parameter saw the loss
            return (None, None, None, None) + tuple(None for _ in params)
        up = {k: g for k, g in zip(_RenderNeuS.KEYS, grads) if g is not None and g.numel() > 0}
        # the fused backward leaves its results in ``.grad``: run it on empty ``.grad`` slots and hand what it produced to
        # autograd (which accumulates, fires DDP / GradScaler hooks ...), then put the previous contents back
        saved = [p.grad for p in params]
        for p in params:
            p.grad = None
        try:
            ctx.step.grad_scale = ctx.grad_scale
            ctx.finish(up)
            # (zero-element parameters -- the SH encoding -- get their empty gradient: DDP waits for one from every parameter)
            out = [p.grad if (p.grad is not None or p.numel() > 0) else torch.zeros_like(p) for p in params]
            ctx.step.release_gradient_buffers()  # (the small gradients are views of per-network buffers: they leave with them)
        finally:
            for p, g in zip(params, saved):
                p.grad = g
        ctx.finish = None
        return (None, None, None, None) + tuple(out)


class FusedNeuSModel(HotPathState):
    """``models.make('neus', config)`` (reference models/neus.py:47-323) on the fused step: NeuS (analytic normals), NeuS with
    the learned NeRF++ background, neuralangelo (progressive levels, finite differences).  Same constructor argument, same
    state-dict keys, same output dict (``num_samples*`` are CPU tensors: the counts are already on the host).  Differentiable
    outputs: comp_rgb_full, comp_rgb, opacity, depth, weights, sdf_samples, sdf_grad_samples, sdf_laplace_samples -- every
    output the reference's system puts a loss on (systems/neus.py:96-139); comp_normal and the background branch's own
    outputs (``*_bg``) carry no gradient of their own (the background trains through comp_rgb_full)."""

    def __init__(self, config):
        cfg = _plain(config)
        if cfg.get("name") != "neus":
            raise ValueError("FusedNeuSModel builds the 'neus' model section")
        super().__init__(cfg)
        self._step, self._last = None, None
        self.refresh_owned_by_trainer = False  # this entry refreshes its grids in update_step, like the reference's model

    def _runner(self):
        if self._step is None:
            from .fused_neus import FusedNeuSStep
            self._step = FusedNeuSStep(self, {})
        return self._step

    # -- systems/base.py:54-57 -> models/neus.py:79-111 -----------------------------------------------------------------
    def update_step(self, epoch, global_step):
        self.restore_schedules(global_step)  # cos anneal, progressive level, finite-difference eps
        cfg = self.config
        if self.training and cfg["grid_prune"] and global_step % 16 == 0:
            step = self._runner()
            step.refresh_occupancy_async(int(global_step), occ_thre=cfg.get("grid_prune_occ_thre", 0.01))
            if step.bg:
                self.occupancy_grid_bg.every_n_step(step=int(global_step), occ_eval_fn=step.bg_occ_eval_fn,
                                                    occ_thre=cfg.get("grid_prune_occ_thre_bg", 0.01))

    def forward_(self, rays):
        bg = self.background_color if self.background_color is not None else torch.ones(3, device=rays.device)
        params = [p for p in self.parameters() if p.requires_grad or p.numel() > 0]
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        full, rgb, opacity, depth, weights, sdf, sdf_grad, lap = _RenderNeuS.apply(self, need_grad, rays, bg, *params)
        last = self._last
        cpu_count = lambda n: torch.as_tensor([int(n)], dtype=torch.int32)  # noqa: E731
        out = {"comp_rgb": rgb, "comp_normal": last["comp_normal"], "opacity": opacity, "depth": depth,
               "rays_valid": opacity > 0, "num_samples": cpu_count(last["num_samples"])}
        if self.training:
            t0, t1 = last["t_starts"].view(-1), last["t_ends"].view(-1)
            out.update({"sdf_samples": sdf, "sdf_grad_samples": sdf_grad, "weights": weights.view(-1),
                        "points": ((t0 + t1) / 2.0), "intervals": (t1 - t0), "ray_indices": last["ray_indices"].view(-1)})
            if self.grad_type == "finite_difference":
                out["sdf_laplace_samples"] = lap
        if self.config.get("learned_background", False):
            out.update({"comp_rgb_bg": last["comp_rgb_bg"], "opacity_bg": last["opacity_bg"], "depth_bg": last["depth_bg"],
                        "rays_valid_bg": last["opacity_bg"] > 0, "num_samples_bg": cpu_count(last["num_samples_bg"])})
            if self.training:
                b0, b1 = last["t_starts_bg"].view(-1), last["t_ends_bg"].view(-1)
                out.update({"weights_bg": last["weights_bg"].view(-1), "points_bg": (b0 + b1) / 2.0, "intervals_bg": b1 - b0,
                            "ray_indices_bg": last["ray_indices_bg"].view(-1)})
            out.update({"comp_rgb_full": full, "num_samples_full": cpu_count(last["num_samples_full"]),
                        "rays_valid_full": (opacity > 0) | (last["opacity_bg"] > 0)})
        else:  # models/neus.py:259-264: a constant background
            out.update({"comp_rgb_bg": bg[None, :].expand(*rgb.shape), "num_samples_bg": torch.zeros(1, dtype=torch.int32),
                        "rays_valid_bg": torch.zeros_like(out["rays_valid"]), "comp_rgb_full": full,
                        "num_samples_full": cpu_count(last["num_samples"]), "rays_valid_full": opacity > 0})
        return out

    def forward(self, rays):
        if self.training:
            out = self.forward_(rays)
            # (every count of this entry is the current forward's -- its host reads them anyway; what is deferred are only the
            # boolean-mask selections, whose losses have the same value: on unless model.defer_mask_selections = False /
            # NSR_BOUNDARY_EAGER)
            if getattr(self, "defer_mask_selections", not os.environ.get("NSR_BOUNDARY_EAGER")):
                # systems/neus.py:98,102 index comp_rgb_full / rgb with rays_valid_full[..., 0]: deferred selections, the MSE /
                # L1 over them as masked means on the device (see _ValidMask / _MaskedLoss above) -- no nonzero(), no wait
                for k in ("rays_valid", "rays_valid_bg", "rays_valid_full"):
                    if k in out and out[k].dtype == torch.bool:
                        out[k] = out[k].as_subclass(_ValidMask)
        else:
            from .export import chunk_batch
            with torch.no_grad():
                out = chunk_batch(self.forward_, int(self.config["ray_chunk"]), True, rays)
        return {**out, "inv_s": torch.exp(self.variance.variance.detach() * 10.0)}

    def eval(self):
        self.randomized = False
        return super().eval()

    def regularizations(self, out):
        return {}  # models/geometry.py / models/texture.py: both regularizations() are empty for these models

    def isosurface(self):
        raise NotImplementedError("marching cubes is a CPU library call in the reference (out of the hot path's scope): "
                                  "nsr.export.isosurface_levels evaluates the level lattice on the device")


def register(models_module):
    """point the reference's registry (models/__init__.py:1-13) at the fused entries"""
    models_module.models["nerf"] = FusedNeRFModel
    models_module.models["neus"] = FusedNeuSModel
    return models_module
