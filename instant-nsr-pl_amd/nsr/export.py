"""Evaluation / export side of the hot path: the same kernels, batched inference.

  * ``render_rays``  -- the ``.eval()`` forward of the reference's models (models/nerf.py:129-139, models/neus.py:289-296):
    rays rendered in ``ray_chunk`` pieces with ``chunk_batch`` semantics (models/utils.py:13-50: per-chunk results detached,
    optionally moved to the CPU, concatenated), no stratified jitter, no occupancy update, white background
    (systems/nerf.py:74-76), through the fused runners with gradients off;
  * ``isosurface_levels`` -- the grid evaluation of ``BaseImplicitGeometry.isosurface_`` (models/geometry.py:83-100): the
    level function (-density for ``volume-density``, sdf for ``volume-sdf``) on a ``resolution^3`` lattice of the box
    [vmin, vmax], evaluated in ``chunk``-point pieces (configs: 512^3 in 2,097,152-point chunks), returned on the CPU as the
    [res, res, res] volume ``mcubes.marching_cubes`` takes.  Marching cubes itself (a CPU library call in the reference) and
    the .obj writer are outside the hot path.
  * ``vertex_colors`` -- the per-vertex colour query of ``export()`` (models/nerf.py:152-161: viewing direction -z;
    models/neus.py:313-323: viewing direction = -normal, the "albedo"), in ``chunk_size`` pieces.
  * checkpoints: ``nsr.state.HotPathState`` has the reference's state-dict keys; ``load_reference_checkpoint`` strips the
    ``model.`` prefix Lightning adds (utils/mixins.py:211-222 saves meshes, Lightning's ModelCheckpoint the weights).
"""
import ctypes

import torch

from nerfacc import ContractionType
from nsr_hip import check, lib, ptr, stream_ptr
from nsr_hip import ops as _ops

_byref = ctypes.byref


def chunk_batch(func, chunk_size, move_to_cpu, *args, **kwargs):
    """models/utils.py:13-50 for dict / tensor / tuple results"""
    B = next(a.shape[0] for a in args if isinstance(a, torch.Tensor))
    out, kind, length = {}, None, 0
    for i in range(0, B, chunk_size):
        o = func(*[a[i:i + chunk_size] if isinstance(a, torch.Tensor) else a for a in args], **kwargs)
        if o is None:
            continue
        kind = type(o)
        if isinstance(o, torch.Tensor):
            o = {0: o}
        elif isinstance(o, (tuple, list)):
            length, o = len(o), dict(enumerate(o))
        for k, v in o.items():
            if not isinstance(v, torch.Tensor):
                continue
            v = v if torch.is_grad_enabled() else v.detach()
            out.setdefault(k, []).append(v.cpu() if move_to_cpu else v)
    if kind is None:
        return None
    out = {k: torch.cat(v, dim=0) for k, v in out.items()}
    if kind is torch.Tensor:
        return out[0]
    if kind in (tuple, list):
        return kind([out[i] for i in range(length)])
    return out


@torch.no_grad()
def render_rays(runner, rays, background=None, chunk=None, move_to_cpu=True):
    """``runner``: FusedNeRFStep or FusedNeuSStep.  -> dict(comp_rgb[_full], opacity, depth, (comp_normal), rays_valid,
    num_samples) like the reference's eval forward"""
    model = runner.model
    chunk = int(chunk or model.config["ray_chunk"])
    dev = rays.device
    bg = torch.ones(3, device=dev) if background is None else background
    was = model.randomized
    model.randomized = False
    keys = ("comp_rgb", "comp_rgb_full", "comp_normal", "opacity", "depth", "rays_valid", "rays_valid_full")

    def one(r):
        gt = torch.zeros((r.shape[0], 3), device=dev)
        if hasattr(runner, "sdf"):  # NeuS runner
            res = runner.forward_backward(r, gt, None, bg, compute_grads=False)
        else:
            res = runner.forward_backward(r, gt, bg, compute_grads=False)
        o = {k: res[k] for k in keys if k in res}
        o["num_samples"] = torch.as_tensor([int(res["num_samples"])], dtype=torch.int32, device=dev)
        return o
    try:
        return chunk_batch(one, chunk, move_to_cpu, rays)
    finally:
        model.randomized = was


def _require_schedules(state):
    """a progressive / finite-difference model straight out of ``load_state_dict`` has no current level and no eps yet (they
    are functions of the global step, restored by the reference's batch-start hooks): refuse to export with a guess"""
    progressive = getattr(state, "progressive", None) is not None
    fd = getattr(state, "grad_type", None) == "finite_difference"
    if (progressive or fd) and not getattr(state, "schedules_restored", True):
        raise RuntimeError("export from a freshly loaded progressive / finite-difference model: call "
                           "state.restore_schedules(global_step) (or load_reference_checkpoint(ckpt, global_step=...)) first "
                           "-- the active level count and the finite-difference eps depend on the training step")


@torch.no_grad()
def forward_level(state, points):
    """models/geometry.py:132-136 (-density) / :212-217 (sdf) on world points [n, 3]"""
    cfg = state.config
    radius = float(cfg["radius"])
    x01 = _ops.contract_to_unisphere(points.float().contiguous(), radius, ContractionType.AABB.value)
    if cfg["name"] == "nerf":
        out = state.geometry.encoding_with_network(x01)
        dens, _ = _ops.density_activation(out.contiguous(), out.shape[1], float(cfg["geometry"].get("density_bias", 0.0)),
                                          want_feature=False)
        return -dens
    from .fused_neus import FusedNeuSStep
    _require_schedules(state)
    runner = getattr(state, "_level_runner", None)
    if runner is None:
        runner = state._level_runner = FusedNeuSStep(state)
    enc = runner.enc
    e = _ops.hashgrid_forward(x01, enc.table_half(enc.params), enc.grid_desc, runner._mask_count())
    blob = runner.sdf.build(requires_grad=False)
    n = x01.shape[0]
    out = torch.empty((n, 16), dtype=torch.float32, device=x01.device)
    with torch.cuda.device(x01.device):
        check(lib.nsr_vmlp_forward(_byref(runner.sdf.desc), ptr(blob), ptr(x01), 3, ptr(e), runner.n_enc, ptr(out), None, None,
                                   n, n, None, stream_ptr()), "nsr_vmlp_forward(level)")
    return out[:, 0]


@torch.no_grad()
def isosurface_levels(state, resolution, vmin=None, vmax=None, chunk=2097152):
    """-> float32 [res, res, res] on the CPU (x slowest, 'ij' meshgrid order of MarchingCubeHelper.grid_vertices)"""
    r = float(state.config["radius"])
    vmin = (-r, -r, -r) if vmin is None else vmin
    vmax = (r, r, r) if vmax is None else vmax
    dev = state.scene_aabb.device
    lin = torch.linspace(0, 1, resolution)
    out = torch.empty(resolution ** 3, dtype=torch.float32)
    per_x = max(1, chunk // (resolution * resolution))  # whole x-slabs per chunk: no 134 M-point lattice in memory
    for i0 in range(0, resolution, per_x):
        xs = lin[i0:i0 + per_x]
        gx, gy, gz = torch.meshgrid(xs, lin, lin, indexing="ij")
        unit = torch.stack([gx.reshape(-1), gy.reshape(-1), gz.reshape(-1)], dim=-1).to(dev)
        pts = torch.stack([unit[:, k] * (vmax[k] - vmin[k]) + vmin[k] for k in range(3)], dim=-1)
        lv = forward_level(state, pts)
        out[i0 * resolution * resolution:(i0 + xs.numel()) * resolution * resolution] = lv.float().cpu()
    return out.view(resolution, resolution, resolution)


@torch.no_grad()
def vertex_colors(state, v_pos, chunk=2097152):
    """``export_vertex_color`` of the reference's ``export()``: colours of mesh vertices ``v_pos`` [n, 3] (world), on the
    CPU.  nerf: ``texture(feature, (0, 0, -1))`` clamped to [0, 1] (models/nerf.py:155-159); neus: ``texture(feature,
    -normal, normal)`` with the normal of the SDF at the vertex (models/neus.py:316-321)."""
    cfg = state.config
    _require_schedules(state)
    dev = state.scene_aabb.device
    if cfg["name"] == "nerf":
        radius = float(cfg["radius"])
        bias = float(cfg["geometry"].get("density_bias", 0.0))
        sh, net = state.texture.encoding.encoding, state.texture.network

        def one(p):
            x01 = _ops.contract_to_unisphere(p.float().contiguous(), radius, ContractionType.AABB.value)
            out = state.geometry.encoding_with_network(x01)
            _, feature = _ops.density_activation(out.contiguous(), out.shape[1], bias, want_feature=True)
            dirs = torch.zeros((p.shape[0], 3), device=p.device)
            dirs[:, 2] = -1.0
            emb = sh((dirs + 1.0) / 2.0)
            rgb = net(torch.cat([feature.to(emb.dtype), emb], dim=-1)).float()[:, :3]  # (output_activation inside the MLP)
            if cfg["texture"].get("color_activation") == "sigmoid":  # models/texture.py:28-29
                rgb = torch.sigmoid(rgb)
            elif "color_activation" in cfg["texture"]:
                raise NotImplementedError("vertex_colors: color_activation " + str(cfg["texture"]["color_activation"]))
            return rgb.clamp(0.0, 1.0)
        return chunk_batch(one, chunk, True, v_pos.to(dev))
    from .fused_neus import FusedNeuSStep
    runner = getattr(state, "_level_runner", None)
    if runner is None:
        runner = state._level_runner = FusedNeuSStep(state)
    return chunk_batch(lambda p: runner.surface_attributes(p)["rgb"], chunk, True, v_pos.to(dev))
