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
A maintainer switches the reference over with one line::

    import nsr.models; nsr.models.register(models)          # models.make('nerf', cfg) now builds FusedNeRFModel

The modular path (the reference's own models/*.py on the drop-in tinycudann / nerfacc packages) stays available; this
entry is the fast one.
"""
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


class _RenderNeRF(torch.autograd.Function):
    """comp_rgb, opacity, depth, weights = render(rays, background; geometry params, texture params)"""

    @staticmethod
    def forward(ctx, model, need_grad, rays, background, p_geometry, p_texture):
        # (need_grad is decided by the caller: grad mode is always off inside Function.forward)
        step = model._runner()
        out, state = step.render_forward(rays.detach(), background.detach(), prepare_backward=need_grad)
        ctx.prepared = bool(need_grad)
        ctx.step, ctx.state = step, state
        model._last = out  # the non-differentiable outputs (ray_indices, t_starts, t_ends, counts) for forward_()
        ctx.mark_non_differentiable(out["ray_indices"])
        return out["comp_rgb"], out["opacity"], out["depth"], out["weights"], out["ray_indices"]

    @staticmethod
    def backward(ctx, g_comp, g_opacity, g_depth, g_weights, _g_ri):
        if not ctx.prepared:
            raise RuntimeError("FusedNeRFModel: backward through a forward that ran without gradients enabled")
        if g_comp is None:  # the loss did not touch the colours: their upstream gradient is zero
            g_comp = torch.zeros((ctx.state["n_rays"], 3), device=ctx.state["ws"].device)
        g1, g2 = ctx.step.render_backward(ctx.state, g_comp, g_opacity, g_depth, g_weights)
        ctx.state = None  # the workspaces go back to the allocator
        return None, None, None, None, g1, g2


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
        comp_rgb, opacity, depth, weights, ray_indices = _RenderNeRF.apply(self, need_grad, rays, bg, ewn.params, tex.params)
        last = self._last
        out = {"comp_rgb": comp_rgb, "opacity": opacity, "depth": depth, "rays_valid": opacity > 0,
               "num_samples": torch.as_tensor([last["num_samples"]], dtype=torch.int32)}
        if self.training:
            t0, t1 = last["t_starts"], last["t_ends"]
            out.update({"weights": weights.view(-1), "points": ((t0 + t1) / 2.0).view(-1), "intervals": (t1 - t0).view(-1),
                        "ray_indices": ray_indices.view(-1)})
        return out

    def forward(self, rays):
        if self.training:
            return {**self.forward_(rays)}
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


def register(models_module):
    """point the reference's registry (models/__init__.py:1-13) at the fused entries"""
    models_module.models["nerf"] = FusedNeRFModel
    return models_module
