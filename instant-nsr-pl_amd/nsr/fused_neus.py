"""Fused NeuS / neuralangelo training step: the computation of ``NeuSModel.forward_`` (reference models/neus.py:205-287)
+ the loss terms of ``NeuSSystem.training_step`` (systems/neus.py:96-130) + backward, issued as ~25 hand-chained kernel
launches instead of ~250 through autograd (hash encode, fp32 SDF network on f32 MFMA with the analytic-normal double
backward or the seven-point finite-difference stencil, SDF->alpha, colour network, alpha compositing, losses).

It reads the parameters of ANY object that exposes them under the reference's attribute paths -- the reference's own
``models.neus.NeuSModel`` on the drop-in packages, ``tests/refmirror`` or an ``nsr.state.HotPathState`` -- and leaves the
gradients in ``.grad`` (same tensors an optimizer would step).  Weight norm of the SDF network is folded on the host:
the effective matrices are formed with a handful of tiny torch ops and the kernel's gradient is pushed back through them.

Scope: configs/neus-blender.yaml (analytic normals, fused fp16 colour MLP), configs/neuralangelo-dtu-wmask.yaml
(progressive levels, finite differences, fp32 colour MLP) and configs/neus-dtu.yaml (fp32 colour MLP + the NeRF++
background of ``forward_bg_``, models/neus.py:169-203: a second hash grid on the contracted space, fp32 density / colour
heads, cone marching through the 256^3 grid with transmittance pruning, density compositing; joined with the foreground
as comp_rgb_full = comp_rgb + comp_rgb_bg (1 - opacity)).
"""
import ctypes
import os

import torch

from nerfacc import ContractionType
from nsr_hip import NsrAdamSegment, NsrVanillaLayer, NsrVmlpDesc, check, device_guard, lib, ptr, stream_ptr
from nsr_hip import shared_stream as _shared_stream
from nsr_hip import ops as _ops

# layout of the encoded features between the encode and the fp32 MLP kernels of a step: 2 = tile-major (default),
# 0 = row-major (developer switch for A/B)
ENC_LAYOUT = int(os.environ.get("NSR_NEUS_ENC_LAYOUT", "2"))
# grid refresh: draw the random cells in increasing order (sorted_uniform_); 0 = unordered (developer switch for A/B)
SORTED_REFRESH = not os.environ.get("NSR_NEUS_UNSORTED_REFRESH")

F32, F16 = torch.float32, torch.float16
_byref = ctypes.byref
LOSS_KEYS = ("lambda_rgb_l1", "lambda_rgb_mse", "lambda_mask", "lambda_opaque", "lambda_eikonal", "lambda_sparsity",
             "lambda_curvature", "sparsity_scale")
ACC = dict(l1=0, mse=1, valid=2, mask=3, opaque=4, eikonal=5, sparsity=6, curvature=7, inv_s_grad=8, rays=9)


def _off(t, n_floats):
    """raw pointer ``n_floats`` fp32 elements into a contiguous tensor (column views of row-major buffers)"""
    return ctypes.c_void_p(t.data_ptr() + 4 * int(n_floats))


def _linear_weight(layer):
    """effective weight of an ``nn.Linear`` with (old-style) weight norm folded: g * v / |v| per output row"""
    if hasattr(layer, "weight_g"):
        v = layer.weight_v
        return layer.weight_g * v / v.norm(dim=1, keepdim=True)
    return layer.weight


class VanillaBlob:
    """parameter blob of csrc/vmlp.hip for the Linear layers of a reference VanillaMLP: ``build`` folds weight norm and
    pads in ONE launch (nsr_vmlp_fold), ``push_gradient`` turns the blob's gradient into ``.grad`` of weight_g / weight_v /
    weight / bias in ONE launch (nsr_vmlp_unfold_gradient) -- the same arithmetic through torch autograd costs ~45
    elementwise / reduction launches per network and step."""

    def __init__(self, layers, n_in, n_out, activation):
        self.layers = list(layers)
        nh = len(self.layers) - 1
        in_pad = {True: 36, False: (n_in + 3) // 4 * 4}[24 < n_in <= 36]  # 36 for the 32/35-input nets (no register spills)
        if in_pad not in (24, 32, 36, 40):
            in_pad = min(p for p in (24, 32, 36, 40) if p >= n_in)
        self.desc = NsrVmlpDesc(int(n_in), int(in_pad), int(n_out), int(nh), int(activation))
        self.n_floats = int(lib.nsr_vmlp_blob_floats(_byref(self.desc)))
        self.blob = None
        self._grad_flat = None

    def _tensors(self, layer):
        wn = hasattr(layer, "weight_g")
        return (layer.weight_v if wn else layer.weight), (layer.weight_g if wn else None), layer.bias

    def _layer_array(self, grads=None):
        arr = (NsrVanillaLayer * len(self.layers))()
        for i, layer in enumerate(self.layers):
            v, g, b = self._tensors(layer)
            for t in (v, g, b):
                if t is not None and (t.dtype != F32 or not t.is_contiguous()):
                    raise NotImplementedError("fused NeuS: VanillaMLP parameters are contiguous fp32 tensors")
            a = arr[i]
            a.weight_v, a.weight_g, a.bias = v.data_ptr(), (None if g is None else g.data_ptr()), b.data_ptr()
            a.n_out, a.n_in = int(v.shape[0]), int(v.shape[1])
            if grads is not None:
                gv, gg, gb = grads[i]
                a.grad_v, a.grad_g, a.grad_bias = gv.data_ptr(), (None if gg is None else gg.data_ptr()), gb.data_ptr()
        return arr

    def build(self, requires_grad=True):
        dev = self.layers[0].bias.device
        if self.blob is None or self.blob.device != dev:
            self.blob = torch.empty(self.n_floats, dtype=F32, device=dev)
        with device_guard(dev):
            check(lib.nsr_vmlp_fold(_byref(self.desc), self._layer_array(), len(self.layers), ptr(self.blob), stream_ptr()),
                  "nsr_vmlp_fold")
        return self.blob

    def push_gradient(self, grad_blob):
        """d loss / d blob -> .grad of weight_g / weight_v / weight / bias (added to gradients that already exist)"""
        params = [t for layer in self.layers for t in self._tensors(layer) if t is not None]
        fresh = all(p.grad is None for p in params)
        if fresh:  # views of one persistent buffer, fully overwritten by the kernel: no allocation, no zeroing
            n = sum(p.numel() for p in params)
            if self._grad_flat is None or self._grad_flat.numel() != n or self._grad_flat.device != grad_blob.device:
                self._grad_flat = torch.empty(n, dtype=F32, device=grad_blob.device)
            off = 0
            for p in params:
                p.grad = self._grad_flat[off:off + p.numel()].view_as(p)
                off += p.numel()
        else:
            for p in params:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
        grads = [tuple(None if t is None else t.grad for t in self._tensors(layer)) for layer in self.layers]
        with device_guard(grad_blob.device):
            check(lib.nsr_vmlp_unfold_gradient(_byref(self.desc), self._layer_array(grads), len(self.layers), ptr(grad_blob),
                                               0 if fresh else 1, stream_ptr()), "nsr_vmlp_unfold_gradient")


def _vanilla_layers(net):
    """the nn.Linear modules of a reference ``VanillaMLP`` (``net.layers`` = Sequential / container: Linear, act, ...)"""
    layers = net.layers
    mods = list(layers.children()) if hasattr(layers, "children") else list(layers)
    return [m for m in mods if hasattr(m, "bias") and (hasattr(m, "weight") or hasattr(m, "weight_v"))]


def sorted_uniform_(buf):
    """fill ``buf`` with n i.i.d. U[0,1) numbers IN INCREASING ORDER (the order statistics of n uniforms are the normalised
    partial sums of n + 1 exponentials): the random cells of a grid refresh are then visited in memory order -- neighbouring
    cells, coherent table gathers in the encode -- and are the same random set in distribution"""
    n = buf.numel()
    if n == 0:
        return buf
    e = torch.empty(n + 1, dtype=torch.float64, device=buf.device).exponential_()
    c = torch.cumsum(e, 0)
    buf.copy_((c[:-1] / c[-1]).clamp_(max=1.0 - 2.0 ** -24))
    return buf


class FusedNeuSStep:
    def __init__(self, model, loss_weights=None):
        cfg = model.config
        if not cfg["grid_prune"]:
            raise NotImplementedError("FusedNeuSStep marches through the occupancy grid(s) (grid_prune: true)")
        self.model = model
        self.bg = bool(cfg["learned_background"])
        self.defer_bg_count = False  # a trainer sets it: its steps read the background's kept count at their end
        if self.bg:
            self._bg_setup(cfg)
        # finite differences: fold the taps that stay in their sample's cell into the sample's table-backward items
        self.fold_taps = not os.environ.get("NSR_FD_PLAIN_TAPS")
        # {"fg": NsrTableAdam, "bg": NsrTableAdam} for ONE step (a trainer sets it): AdamW on the hash tables inside their
        # backward; adam_applied says which of them a step really updated (a step without samples launches no table backward)
        self.table_adam, self.adam_applied = None, set()
        # multi-GPU: {"fg": bf16 send buffer, "bg": ...} of nsr.parallel.ShardedAdamW -- the table backward writes the
        # exchange's transport format itself (no fp32 gradient, no cast); bf16_written: which of them a step really filled
        # scale of dL/dy in front of the fp16 rounding of the fused colour MLP's backward (see nsr/fused.py: tcnn's 128 sits on top of Lightning's GradScaler(65536); this step has no scaler)
        self.grad_scale = float(os.environ.get("NSR_GRAD_SCALE", "65536"))
        self.table_bf16, self.bf16_written = None, set()
        self.grad_written = set()  # tables whose fp32 .grad this step wrote (or cleared): the others' is a previous step's
        self.radius = float(cfg["radius"])
        g = cfg["geometry"]
        self.fd = g["grad_type"] == "finite_difference"
        self.n_feat = int(g["feature_dim"])
        node = model.geometry.encoding.encoding
        self.enc = node if hasattr(node, "grid_desc") else node.encoding  # tcnn.Encoding (HashGrid)
        self._pg = None if hasattr(node, "grid_desc") else node           # ProgressiveBandHashGrid-like holder
        if not g["xyz_encoding_config"].get("include_xyz", False):
            raise NotImplementedError("fused NeuS: the SDF network input is [xyz | hash encoding] (include_xyz: true)")
        d = self.enc.grid_desc
        self.n_enc = int(d.n_levels * d.n_features)
        if self.n_enc + 3 > 40:
            raise NotImplementedError("fused NeuS: the SDF network input (3 + levels x features) is limited to 40 columns")
        sdf_layers = _vanilla_layers(model.geometry.network)
        if len(sdf_layers) != 2:
            raise NotImplementedError("fused NeuS: the SDF network has one hidden layer (every reference config)")
        # options of the reference's modules that no shipped YAML switches on are REFUSED, not silently ignored: a model
        # configured that way would train / render something else (models/neus.py:27-44, models/geometry.py:171-174,
        # models/network_utils.py:133-139)
        if (cfg.get("variance") or {}).get("modulate", False):
            raise NotImplementedError("fused NeuS: variance.modulate (the clamp schedule on inv_s, models/neus.py:27-44) is "
                                      "not implemented -- every reference config sets modulate: false")
        if not g["mlp_network_config"].get("sphere_init", False):
            raise NotImplementedError("fused NeuS: the SDF network is the sphere-initialised Softplus(beta=100) MLP "
                                      "(mlp_network_config.sphere_init: true); the ReLU variant is not compiled")
        for key in ("sdf_activation", "feature_activation"):
            if g.get(key) not in (None, "none", "None"):
                raise NotImplementedError(f"fused NeuS: geometry.{key} is not implemented (no reference config sets it)")
        self.sdf = VanillaBlob(sdf_layers, 3 + self.n_enc, self.n_feat, activation=1)  # softplus(beta=100): sphere_init
        tex = model.texture.network
        self.tex_fused = hasattr(tex, "mlp_desc")
        self.tex = tex if self.tex_fused else VanillaBlob(_vanilla_layers(tex), self.n_feat + 19, 3, activation=0)
        if cfg["texture"].get("color_activation") != "sigmoid":
            raise NotImplementedError("fused NeuS: color_activation sigmoid")
        lw = dict(lambda_rgb_l1=1.0, lambda_eikonal=0.1, lambda_mask=0.1, sparsity_scale=1.0)
        lw.update(loss_weights or {})
        self.loss_weights = lw

    # ---- schedules that live on the model object -------------------------------------------------------------------
    def _mask_count(self):
        if self._pg is None:
            return self.enc.grid_desc.n_levels
        return int(getattr(self._pg, "current_level", getattr(self.model, "current_level", self.enc.grid_desc.n_levels)))

    def _fd_eps(self):
        geo = self.model.geometry
        eps = getattr(geo, "_finite_difference_eps", None)
        if eps is None:
            eps = getattr(self.model, "finite_difference_eps", None)
        return float(eps)

    def _inv_s(self):
        """exp(10 variance) (models/neus.py:27-32) into a persistent one-float buffer"""
        var = self.model.variance.variance
        if getattr(self, "_inv_s_buf", None) is None or self._inv_s_buf.device != var.device:
            self._inv_s_buf = torch.empty(1, dtype=F32, device=var.device)
        with device_guard(var.device):
            check(lib.nsr_neus_inv_s(ptr(var.detach()), ptr(self._inv_s_buf), stream_ptr()), "nsr_neus_inv_s")
        return self._inv_s_buf

    def loss_terms(self, acc):
        """the system's scalar losses from the accumulator (device tensors)"""
        n_s = max(self._n_samples, 1)
        valid, rays = torch.clamp(acc[ACC["valid"]], min=1.0), torch.clamp(acc[ACC["rays"]], min=1.0)
        return {"rgb_l1": acc[ACC["l1"]] / (3.0 * valid), "rgb_mse": acc[ACC["mse"]] / (3.0 * valid),
                "mask": acc[ACC["mask"]] / rays, "opaque": acc[ACC["opaque"]] / rays,
                "eikonal": acc[ACC["eikonal"]] / n_s, "sparsity": acc[ACC["sparsity"]] / n_s,
                "curvature": acc[ACC["curvature"]] / n_s}

    def loss_value(self, acc):
        t = self.loss_terms(acc)
        return sum(float(self.loss_weights.get("lambda_" + k, 0.0)) * v for k, v in t.items())

    # ---- the step ----------------------------------------------------------------------------------------------------
    def march_begin(self, rays_o, rays_d, t_min=None, t_max=None):
        """models/neus.py:209-220 (ray_marching(scene_aabb, grid, alpha_fn=None, stratified=randomized)) queued on the
        CURRENT stream without a host sync -- a trainer runs it for the next batch on a side stream"""
        m, grid = self.model, self.model.occupancy_grid
        if t_min is None:
            t_min, t_max = _ops.ray_aabb_intersect(rays_o, rays_d, m.scene_aabb)
            if m.randomized:
                t_min = t_min + torch.rand_like(t_min) * m.render_step_size
        return _ops.ray_march_begin(rays_o, rays_d, t_min, t_max, grid.roi_aabb, grid.binary, ContractionType.AABB.value,
                                    m.render_step_size, 0.0, roi_host=grid._roi_host)

    def occ_eval_fn(self, x):
        """occupancy statistic of models/neus.py:90-101 (closed-form alpha of one step at a flat SDF) on the kernels"""
        m, enc = self.model, self.enc
        n = x.shape[0]
        with torch.no_grad(), device_guard(x.device):
            x01 = _ops.contract_to_unisphere(x.float().contiguous(), self.radius, ContractionType.AABB.value)
            e = _ops.hashgrid_forward(x01, enc.table_half(enc.params), enc.grid_desc, self._mask_count())
            blob = self.sdf.build(requires_grad=False)
            out = torch.empty((n, 16), dtype=F32, device=x.device)
            check(lib.nsr_vmlp_forward(_byref(self.sdf.desc), ptr(blob), ptr(x01), 3, ptr(e), self.n_enc, ptr(out), None,
                                       None, n, n, None, stream_ptr()), "nsr_vmlp_forward(occupancy)")
            sdf = out[:, :1]
            inv_s = self._inv_s().clip(1e-6, 1e6)
            h = m.render_step_size * 0.5
            prev_cdf, next_cdf = torch.sigmoid((sdf + h) * inv_s), torch.sigmoid((sdf - h) * inv_s)
            return ((prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)).clip(0.0, 1.0)


    # ---- NeRF++ background (models/neus.py:169-203) -----------------------------------------------------------------
    def _bg_setup(self, cfg):
        m, g, t = self.model, cfg["geometry_bg"], cfg["texture_bg"]
        ewn = m.geometry_bg.encoding_with_network
        node = ewn.encoding.encoding if hasattr(ewn, "encoding") else None
        if node is None or not hasattr(node, "grid_desc") or g["xyz_encoding_config"].get("include_xyz", False):
            raise NotImplementedError("fused NeuS background: HashGrid encoding (no xyz columns) + VanillaMLP density head")
        if g.get("density_activation") != "trunc_exp" or "feature_activation" in g:
            raise NotImplementedError("fused NeuS background: density_activation trunc_exp, no feature activation")
        self.bg_enc = node
        d = node.grid_desc
        self.bg_n_enc = int(d.n_levels * d.n_features)
        self.bg_n_feat = int(g["feature_dim"])
        self.bg_bias = float(g.get("density_bias", 0.0))
        layers = _vanilla_layers(ewn.network)
        if len(layers) != 2 or self.bg_n_enc > 40 or self.bg_n_feat > 16:
            raise NotImplementedError("fused NeuS background: density head = VanillaMLP with one hidden layer")
        self.bg_geo = VanillaBlob(layers, self.bg_n_enc, self.bg_n_feat, activation=0)
        tex = m.texture_bg.network
        if hasattr(tex, "mlp_desc") or t.get("color_activation") != "sigmoid":
            raise NotImplementedError("fused NeuS background: fp32 VanillaMLP colour head with sigmoid")
        self.bg_tex = VanillaBlob(_vanilla_layers(tex), self.bg_n_feat + 16, 3, activation=0)
        self.bg_tex_stride = int(self.bg_tex.desc.in_pad)

    def bg_march_begin(self, rays_o, rays_d, t_max=None):
        """ray_marching of forward_bg_ queued without a host sync: near plane = exit of the foreground box (0.1 for rays
        that miss it), far plane 1e3, cone-angle step growth, 256^3 grid on the contracted space"""
        m, grid = self.model, self.model.occupancy_grid_bg
        if t_max is None:
            _, t_max = _ops.ray_aabb_intersect(rays_o, rays_d, m.scene_aabb)
        near = torch.where(t_max > 1e9, torch.full_like(t_max, float(m.near_plane_bg)), t_max)
        if m.randomized:
            near = near + torch.rand_like(near) * float(m.render_step_size_bg)
        far = torch.full_like(near, float(m.far_plane_bg))
        return _ops.ray_march_begin(rays_o, rays_d, near.contiguous(), far, grid.roi_aabb, grid.binary,
                                    grid.contraction_type.value, float(m.render_step_size_bg), float(m.cone_angle_bg),
                                    roi_host=getattr(grid, "_roi_host", None))

    def bg_occ_eval_fn(self, x):
        """occupancy statistic of the background grid (models/neus.py:103-106): density x step size"""
        m, enc = self.model, self.bg_enc
        n = x.shape[0]
        with torch.no_grad(), device_guard(x.device):
            x01 = _ops.contract_to_unisphere(x.float().contiguous(), self.radius, ContractionType.UN_BOUNDED_SPHERE.value)
            e = _ops.hashgrid_forward(x01, enc.table_half(enc.params), enc.grid_desc).float()
            blob = self.bg_geo.build(requires_grad=False)
            out = torch.empty((n, 16), dtype=F32, device=x.device)
            check(lib.nsr_vmlp_forward(_byref(self.bg_geo.desc), ptr(blob), ptr(e), self.bg_n_enc, None, 0, ptr(out), None,
                                       None, n, n, None, stream_ptr()), "nsr_vmlp_forward(bg occupancy)")
            return torch.exp(out[:, :1] + self.bg_bias) * float(m.render_step_size_bg)

    def _bg_prune_begin(self, handle):
        """sigma_fn pass of ray_marching: encode + density head on every marched sample, each ray's leading samples with
        transmittance >= 1e-4 counted and the total sent to the host -- queued without waiting for it"""
        m, enc = self.model, self.bg_enc
        rays_o, rays_d = handle.args[0], handle.args[1]
        dev, n_rays = rays_o.device, rays_o.shape[0]
        pk_m, ri_m, t0_m, t1_m = _ops.ray_march_finish(handle)
        M = ri_m.shape[0]
        s = stream_ptr()
        x01_m = torch.empty((M, 3), dtype=F32, device=dev)
        check(lib.nsr_sample_positions_unit(ptr(rays_o), ptr(rays_d), ptr(ri_m), ptr(t0_m), ptr(t1_m), self.radius,
                                            ContractionType.UN_BOUNDED_SPHERE.value, ptr(x01_m), None, M, None, s),
              "nsr_sample_positions_unit")
        table = enc.table_half(enc.params)
        xin_m = _ops.hashgrid_forward(x01_m, table, enc.grid_desc).float() if M else torch.empty((0, self.bg_n_enc), dtype=F32, device=dev)
        self._bg_blob_g = self.bg_geo.build(requires_grad=self._bg_grads)
        out_m = torch.empty((M, 16), dtype=F32, device=dev)
        check(lib.nsr_vmlp_forward(_byref(self.bg_geo.desc), ptr(self._bg_blob_g.detach()), ptr(xin_m), self.bg_n_enc, None, 0,
                                   ptr(out_m), None, None, M, M, None, s), "nsr_vmlp_forward(bg density, marched)")
        kept = torch.empty(n_rays, dtype=torch.int32, device=dev)
        pk = torch.empty((n_rays, 2), dtype=torch.int32, device=dev)
        total = torch.zeros(1, dtype=torch.int32, device=dev)
        check(lib.nsr_bg_visibility_prefix(ptr(out_m), self.bg_bias, ptr(t0_m), ptr(t1_m), ptr(pk_m), 1e-4, ptr(kept),
                                           n_rays, s), "nsr_bg_visibility_prefix")
        check(lib.nsr_pack_from_counts(ptr(kept), ptr(pk), ptr(total), n_rays, s), "nsr_pack_from_counts")
        return dict(packed=pk, M=M, rays_d=rays_d, n_rays=n_rays, _count=_ops.read_count_begin(total), _total=total,
                    _marched=(pk_m, [t0_m, t1_m, x01_m, xin_m, out_m]))

    def _bg_prune_finish(self, c):
        """the kept sample count reaches the host (the branch's second host sync; the caller has queued the foreground's encode
        and SDF network behind the pruning pass by now, so the GPU works while the host waits -- done right after the pruning
        pass, the wait and the host's queueing after it left the main stream idle ~0.1 ms per step) -> the kept samples' arrays"""
        c["S"] = _ops.read_count_finish(c.pop("_count"))
        return self._bg_gather_kept(c, c["S"], None)

    def _bg_prune_defer(self, c):
        """... or the count does NOT reach the host here (a trainer's own step, ``defer_bg_count``): the kept arrays get a row per
        MARCHED sample, every kernel of the branch takes the kept count from the device (``n_dev``), and the trainer reads it
        when the whole step is queued (``bg_count_deferred``).  With the branch on its own stream the host is what a step waits
        for: a wait for the GPU in the middle of the forward leaves both streams idle while the host catches up afterwards."""
        c["S"] = None
        return self._bg_gather_kept(c, c["M"], c["_total"])

    def bg_count_deferred(self, res):
        """end of a step whose background count was deferred: the count (long on its way) -> ``res``; the per-sample outputs
        are cut to it.  -> S"""
        c, hook = res.pop("_bg_deferred")
        if hook is not None:
            hook(block=True)  # (a no-op when one of the step's polls has already queued the next batch)
        elif "_count" in c:
            c["S"] = _ops.read_count_finish(c.pop("_count"))
        S = c["S"]
        res["num_samples_bg"], res["num_samples_full"] = S, res["num_samples"] + S
        for k in ("weights_bg", "ray_indices_bg", "t_starts_bg", "t_ends_bg"):
            res[k] = res[k][:S]
        return S

    def _bg_gather_kept(self, c, rows, n_dev):
        """the kept samples' arrays (``rows`` rows) <- each ray's leading samples of the marched arrays"""
        dev, n_rays = c["rays_d"].device, c["n_rays"]
        pk_m, srcs = c.pop("_marched")
        R = max(rows, 1)  # (nothing kept -- e.g. an empty background grid: the ray kernels still run, on non-NULL arrays)
        c.update(n=rows, n_dev=n_dev, ri=torch.empty(R, dtype=torch.int64, device=dev)[:rows],
                 t0=torch.empty(R, dtype=F32, device=dev)[:rows], t1=torch.empty(R, dtype=F32, device=dev)[:rows],
                 x01=torch.empty((R, 3), dtype=F32, device=dev)[:rows],
                 xin=torch.empty((R, self.bg_n_enc), dtype=F32, device=dev)[:rows],
                 out=torch.empty((R, 16), dtype=F32, device=dev)[:rows])
        if rows == 0:
            return c
        dsts = [c["t0"], c["t1"], c["x01"], c["xin"], c["out"]]
        k = len(srcs)
        sp = (ctypes.c_void_p * k)(*[t.data_ptr() for t in srcs])
        dp = (ctypes.c_void_p * k)(*[t.data_ptr() for t in dsts])
        rb = (ctypes.c_uint32 * k)(*[(t.stride(0) if t.dim() > 1 else 1) * t.element_size() for t in srcs])
        check(lib.nsr_copy_ray_prefix_rows(ptr(pk_m), ptr(c["packed"]), k, sp, dp, rb, ptr(c["rays_d"]), None, ptr(c["ri"]),
                                           n_rays, stream_ptr()), "nsr_copy_ray_prefix_rows")
        return c

    def _bg_forward(self, c, background):
        """density / colour heads on the kept samples and density compositing -> comp_rgb_bg, opacity_bg"""
        # S rows (the kept count on the host) or one row per marched sample + the kept count on the device (_bg_prune_defer)
        dev, S, nd, n_rays, s = c["x01"].device, c["n"], ptr(c["n_dev"]), c["n_rays"], stream_ptr()
        st = self.bg_tex_stride
        c["tex_in"] = torch.empty((S, st), dtype=F32, device=dev)
        check(lib.nsr_bg_texture_input(ptr(c["out"]), self.bg_n_feat, ptr(c["rays_d"]), ptr(c["ri"]), ptr(c["tex_in"]), st, S,
                                       nd, s), "nsr_bg_texture_input")
        c["rgb_raw"] = torch.empty((S, 16), dtype=F32, device=dev)
        check(lib.nsr_vmlp_forward(_byref(self.bg_tex.desc), ptr(self._bg_blob_t.detach()), ptr(c["tex_in"]), st, None, 0,
                                   ptr(c["rgb_raw"]), None, None, S, S, nd, s), "nsr_vmlp_forward(bg colour)")
        c["weights"], c["trans"] = torch.empty(S, dtype=F32, device=dev), torch.empty(S, dtype=F32, device=dev)
        c["comp_rgb"] = torch.empty((n_rays, 3), dtype=F32, device=dev)
        c["opacity"] = torch.empty((n_rays, 1), dtype=F32, device=dev)
        c["depth"] = torch.empty((n_rays, 1), dtype=F32, device=dev)
        c["bg_colour"] = background
        check(lib.nsr_bg_composite_forward(ptr(c["packed"]), ptr(c["out"]), self.bg_bias, ptr(c["rgb_raw"]), ptr(c["t0"]),
                                           ptr(c["t1"]), ptr(background), ptr(c["weights"]), ptr(c["trans"]),
                                           ptr(c["comp_rgb"]), ptr(c["opacity"]), ptr(c["depth"]), n_rays, s),
              "nsr_bg_composite_forward")
        # bin the kept samples for the table backward underneath the foreground's compositing / backward (queued after the
        # branch's own forward kernels: its host-side setup must not hold those back)
        if self._bg_grads and S > 0:
            desc = self.bg_enc.grid_desc
            c["gws"] = torch.empty(int(lib.nsr_hashgrid_backward_params_workspace_floats(_byref(desc), S)), dtype=F32, device=dev)
            c["bin_event"] = self._on_helper(lambda hs: check(lib.nsr_hashgrid_backward_params_owner_bin(
                ptr(c["x01"]), ptr(c["gws"]), S, desc.n_levels, _byref(desc), nd, hs),
                "nsr_hashgrid_backward_params_owner_bin(bg)"), (c["gws"], c["x01"]))

    def _bg_backward(self, c, d_comp):
        """d comp_rgb_bg [n_rays, 3] -> gradients of the background's table / density head / colour head.
        -> (g_geo blob gradient, g_tex blob gradient) for the host-side push"""
        dev, S, nd, n_rays, s = c["x01"].device, c["n"], ptr(c["n_dev"]), c["n_rays"], stream_ptr()
        enc, desc, st = self.bg_enc, self.bg_enc.grid_desc, self.bg_tex_stride
        gd, td = self.bg_geo.desc, self.bg_tex.desc
        d_logit = torch.empty(S, dtype=F32, device=dev)
        d_rgb = torch.empty((S, 16), dtype=F32, device=dev)
        check(lib.nsr_bg_composite_backward(ptr(c["packed"]), ptr(c["out"]), self.bg_bias, ptr(c["rgb_raw"]),
                                            ptr(c["weights"]), ptr(c["trans"]), ptr(c["t0"]), ptr(c["t1"]),
                                            ptr(c["bg_colour"]), ptr(d_comp), ptr(d_logit), ptr(d_rgb), n_rays, s),
              "nsr_bg_composite_backward")
        d_tex = torch.empty((S, st), dtype=F32, device=dev)
        g_tex = torch.empty(self.bg_tex.n_floats, dtype=F32, device=dev)
        ws = torch.empty(int(lib.nsr_vmlp_backward_workspace_floats(_byref(td), S)), dtype=F32, device=dev)
        check(lib.nsr_vmlp_backward(_byref(td), ptr(self._bg_blob_t.detach()), ptr(c["tex_in"]), st, None, 0, ptr(d_rgb), None,
                                    None, ptr(d_tex), st, 0, st, 0, ptr(g_tex), 0, ptr(ws), S, S, nd, s),
              "nsr_vmlp_backward(bg colour)")
        d_out = torch.empty((S, 16), dtype=F32, device=dev)
        check(lib.nsr_bg_join_gradients(ptr(d_logit), ptr(d_tex), st, self.bg_n_feat, ptr(d_out), S, nd, s),
              "nsr_bg_join_gradients")
        C, F = self.bg_n_enc, int(desc.n_features)
        d_enc = torch.empty(C * S, dtype=F32, device=dev)  # level-major: what the owner-computes table backward reads
        g_geo = torch.empty(self.bg_geo.n_floats, dtype=F32, device=dev)
        ws2 = torch.empty(int(lib.nsr_vmlp_backward_workspace_floats(_byref(gd), S)), dtype=F32, device=dev)
        check(lib.nsr_vmlp_backward(_byref(gd), ptr(self._bg_blob_g.detach()), ptr(c["xin"]), C, None, 0, ptr(d_out), None, None,
                                    ptr(d_enc), 0, 0, C, F, ptr(g_geo), 0, ptr(ws2), S, S, nd, s),
              "nsr_vmlp_backward(bg density)")
        if enc.params.grad is None:
            enc.params.grad = torch.zeros_like(enc.params)
        ad = (self.table_adam or {}).get("bg")
        bf = None if ad is not None else (self.table_bf16 or {}).get("bg")
        if S > 0 and bf is not None:
            torch.cuda.current_stream().wait_event(c["bin_event"])
            check(lib.nsr_hashgrid_backward_params_owner_accumulate_range(
                ptr(c["x01"]), ptr(d_enc), None, ptr(bf), ptr(c["gws"]), S, desc.n_levels, 1.0, 0, desc.n_levels, _byref(desc),
                nd, s), "nsr_hashgrid_backward_params_owner_accumulate_range(bg)")
            self.bf16_written.add("bg")
        elif S > 0 and ad is not None:
            torch.cuda.current_stream().wait_event(c["bin_event"])
            check(lib.nsr_hashgrid_backward_params_owner_accumulate_adam(ptr(c["x01"]), ptr(d_enc), 2, 0, ptr(c["gws"]), S,
                                                                         desc.n_levels, 1.0, _byref(desc), nd, _byref(ad), s),
                  "nsr_hashgrid_backward_params_owner_accumulate_adam(bg)")
            self.adam_applied.add("bg")
        elif S > 0:
            torch.cuda.current_stream().wait_event(c["bin_event"])
            check(lib.nsr_hashgrid_backward_params_owner_accumulate(ptr(c["x01"]), ptr(d_enc), 2, 0, ptr(enc.params.grad),
                                                                    ptr(c["gws"]), S, desc.n_levels, 1.0, 0, _byref(desc),
                                                                    nd, s),
                  "nsr_hashgrid_backward_params_owner_accumulate(bg)")
            self.grad_written.add("bg")
        else:
            enc.params.grad.zero_()
            self.grad_written.add("bg")
        return g_geo, g_tex

    def _finish_bg_only(self, res, g_bg):
        with torch.enable_grad():
            self.bg_geo.push_gradient(g_bg[0])
            self.bg_tex.push_gradient(g_bg[1])
        return res

    def _bg_ctx(self, dev, behind_main=True, behind=None):
        """``with self._bg_ctx(dev):`` -- the background branch's kernels go to their own stream, behind everything the main
        stream has queued so far (``behind_main``).  The branch is ~25 launches of 5-90 us on 1e5 / 4e4 samples; on the main
        stream they sat in front of the foreground's encode (pruning pass) and colour backward (its backward): 0.4 ms of a 2.2
        ms step that now runs beside the foreground's networks.  ``NSR_NEUS_BG_ON_MAIN``: the old order (A/B switch)."""
        import contextlib
        if os.environ.get("NSR_NEUS_BG_ON_MAIN"):
            return contextlib.nullcontext()
        if getattr(self, "_bg_stream", None) is None:
            self._bg_stream = _shared_stream(dev, "bg")
        if behind is not None:  # (an event of the main stream recorded earlier: e.g. the start of the step)
            self._bg_stream.wait_event(behind)
        elif behind_main:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self._bg_stream.wait_event(ev)
        return torch.cuda.stream(self._bg_stream)

    def _bg_join(self, tensors):
        """the current (main) stream waits for the background stream; ``tensors`` were allocated there and are used here"""
        bgs = getattr(self, "_bg_stream", None)
        if bgs is None or os.environ.get("NSR_NEUS_BG_ON_MAIN"):
            return
        ev = torch.cuda.Event()
        ev.record(bgs)
        cur = torch.cuda.current_stream()
        cur.wait_event(ev)
        for t in tensors:
            if isinstance(t, torch.Tensor):
                t.record_stream(cur)

    def _bg_uses(self, *tensors):
        """tensors of the main stream that the background stream's kernels read"""
        bgs = getattr(self, "_bg_stream", None)
        if bgs is not None and not os.environ.get("NSR_NEUS_BG_ON_MAIN"):
            for t in tensors:
                if isinstance(t, torch.Tensor):
                    t.record_stream(bgs)

    def _on_helper(self, fn, tensors):
        """run ``fn(stream_ptr)`` on the helper stream behind everything queued so far; -> completion event"""
        dev = tensors[0].device
        if getattr(self, "_helper", None) is None:
            self._helper = _shared_stream(dev, "helper")
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        self._helper.wait_event(ready)
        with torch.cuda.stream(self._helper):
            fn(stream_ptr())
            ev = torch.cuda.Event()
            ev.record(self._helper)
        for t in tensors:
            t.record_stream(self._helper)
        return ev

    @torch.no_grad()
    def refresh_occupancy_async(self, step, occ_thre=0.01, ema_decay=0.95, warmup_steps=256):
        """``OccupancyGrid._update`` of the FOREGROUND grid (nerfacc 0.3.3; reference models/neus.py:79-109) queued on the
        current stream with the selected-cell count kept on the device (csrc/occupancy.hip): cell selection, positions,
        encode, SDF network, closed-form alpha, EMA / threshold / binarise, brick re-packing -- ~14 launches and no host
        synchronisation instead of the ~250 launches + ``torch.nonzero`` of the torch formulation."""
        m, grid, enc, desc = self.model, self.model.occupancy_grid, self.enc, self.enc.grid_desc
        dev = grid.occs.device
        rx, ry, rz = grid._res
        N = grid.num_cells
        all_cells = step < warmup_steps
        n_uniform = N // 4
        cap = N if all_cells else 2 * n_uniform
        ob = getattr(self, "_occ_buf", None)
        binary = grid._binary
        assert binary.is_contiguous() and binary.dtype == torch.bool
        if ob is None:
            bricks = torch.empty(int(lib.nsr_grid_bricks_words64(rx, ry, rz)), dtype=torch.int64, device=dev)
            ob = self._occ_buf = dict(
                bricks=bricks, cells=torch.empty(N, dtype=torch.int32, device=dev),
                x_unit=torch.empty(N * 3, dtype=F32, device=dev), world=torch.empty(N * 3, dtype=F32, device=dev),
                x01=torch.empty(N * 3, dtype=F32, device=dev), enc=torch.empty(N * self.n_enc, dtype=F16, device=dev),
                out=torch.empty(N * 16, dtype=F32, device=dev), occ=torch.empty(N, dtype=F32, device=dev),
                u=torch.empty(2 * n_uniform, dtype=F32, device=dev), jitter=torch.empty(N * 3, dtype=F32, device=dev),
                brick_offset=torch.empty(max(bricks.numel(), 1), dtype=torch.int32, device=dev),
                occupied=torch.empty(N, dtype=torch.int32, device=dev), counts=torch.zeros(4, dtype=torch.int32, device=dev),
                occs=torch.empty(N, dtype=F32, device=dev), thr=torch.empty(2 + 2 * 256, dtype=F32, device=dev))
        bricks = ob["bricks"]
        _ops.grid_bricks(binary, out=bricks)  # the bitfield of the CURRENT grid (cached per tensor version)
        n_occ, n_cells = ob["counts"][0:1], ob["counts"][1:2]
        table = enc.table_half(enc.params)
        blob = self.sdf.build(requires_grad=False)
        inv_s = self._inv_s()
        with device_guard(dev):
            s = stream_ptr()
            ob["jitter"][:cap * 3].uniform_()
            if not all_cells:
                if SORTED_REFRESH:
                    sorted_uniform_(ob["u"][:n_uniform])
                    sorted_uniform_(ob["u"][n_uniform:])
                else:
                    ob["u"].uniform_()
            check(lib.nsr_occupancy_select_cells(ptr(bricks), rx, ry, rz, ptr(ob["u"][:n_uniform]),
                                                 ptr(ob["u"][n_uniform:]), ptr(ob["jitter"]), n_uniform, int(all_cells),
                                                 cap, ptr(ob["brick_offset"]), ptr(ob["occupied"]), ptr(n_occ),
                                                 ptr(ob["cells"]), ptr(ob["x_unit"]), ptr(n_cells), s),
                  "nsr_occupancy_select_cells")
            check(lib.nsr_contract_inv(ptr(ob["x_unit"]), ptr(grid.roi_aabb), ContractionType.AABB.value, ptr(ob["world"]),
                                       cap, s), "nsr_contract_inv")
            check(lib.nsr_contract_to_unisphere(ptr(ob["world"]), self.radius, ContractionType.AABB.value, ptr(ob["x01"]),
                                                cap, s), "nsr_contract_to_unisphere")
            check(lib.nsr_hashgrid_forward_ex(ptr(ob["x01"]), ptr(table), ptr(ob["enc"]), cap, self.n_enc, 0,
                                              self._mask_count(), _byref(desc), ptr(n_cells), s), "nsr_hashgrid_forward_ex")
            check(lib.nsr_vmlp_forward(_byref(self.sdf.desc), ptr(blob), ptr(ob["x01"]), 3, ptr(ob["enc"]), self.n_enc,
                                       ptr(ob["out"]), None, None, cap, cap, ptr(n_cells), s), "nsr_vmlp_forward(occupancy)")
            check(lib.nsr_neus_occupancy_values(ptr(ob["out"]), ptr(inv_s), float(m.render_step_size), ptr(ob["occ"]), cap,
                                                ptr(n_cells), s), "nsr_neus_occupancy_values")
            check(lib.nsr_occupancy_update_values(ptr(ob["occ"]), float(ema_decay), float(occ_thre), ptr(ob["cells"]),
                                                  ptr(grid.occs), ptr(ob["occs"]), ptr(binary.view(torch.uint8)),
                                                  ptr(ob["thr"]), N, cap, ptr(n_cells), s), "nsr_occupancy_update_values")
            grid.occs.copy_(ob["occs"])
            check(lib.nsr_grid_pack_bricks(ptr(binary.view(torch.uint8)), rx, ry, rz, ptr(bricks), s),
                  "nsr_grid_pack_bricks")
        try:  # the cache of ops.grid_bricks keys on the tensor version, which a raw-pointer write does not bump
            binary._nsr_bricks = (binary._version, binary.data_ptr(), bricks)
        except Exception:  # noqa: BLE001
            pass

    @torch.no_grad()
    def refresh_bg_occupancy_async(self, step, occ_thre=0.01, ema_decay=0.95, warmup_steps=256):
        """``OccupancyGrid._update`` of the 256^3 NeRF++ BACKGROUND grid (nerfacc 0.3.3, UN_BOUNDED_SPHERE; reference
        models/neus.py:103-111) on the current stream, the selected-cell count kept on the device: cell selection, positions,
        encode (tile-major fp16), the fp32 density head reading the encoding directly, exp(logit + bias) * step inside the unit
        sphere, EMA / threshold / binarise, brick re-packing -- 11 launches.  The torch formulation of the same refresh
        (nerfacc/grid.py with ``bg_occ_eval_fn``) spent more than half of its 5.5-7 ms in ~80 torch launches around the encode:
        int64 cell coordinates, boolean-mask compaction, a row-major fp16 encoding (ten times its size in fabric writes)
        converted to fp32 for the network."""
        m, grid, enc, desc = self.model, self.model.occupancy_grid_bg, self.bg_enc, self.bg_enc.grid_desc
        if self.bg_geo.desc.in_pad != 36 or self.bg_n_enc != 32 or int(desc.n_features) != 2:
            raise NotImplementedError("device-side background refresh: 16 levels x 2 features -> one hidden layer")
        dev = grid.occs.device
        rx, ry, rz = grid._res
        N = grid.num_cells
        all_cells = step < warmup_steps
        n_uniform = N // 4
        cap = N if all_cells else 2 * n_uniform
        rows = (cap + 15) // 16 * 16
        ob = getattr(self, "_occ_buf_bg", None)
        binary = grid._binary
        assert binary.is_contiguous() and binary.dtype == torch.bool
        if ob is None or ob["rows"] < rows:
            bricks = torch.empty(int(lib.nsr_grid_bricks_words64(rx, ry, rz)), dtype=torch.int64, device=dev)
            ob = self._occ_buf_bg = dict(
                rows=rows, bricks=bricks, cells=torch.empty(cap, dtype=torch.int32, device=dev),
                x_unit=torch.empty(cap * 3, dtype=F32, device=dev), world=torch.empty(cap * 3, dtype=F32, device=dev),
                x01=torch.empty(cap * 3, dtype=F32, device=dev), enc=torch.empty(rows * self.bg_n_enc, dtype=F16, device=dev),
                logit=torch.empty(cap, dtype=F32, device=dev), occ=torch.empty(cap, dtype=F32, device=dev),
                u=torch.empty(2 * n_uniform, dtype=F32, device=dev), jitter=torch.empty(cap * 3, dtype=F32, device=dev),
                brick_offset=torch.empty(max(bricks.numel(), 1), dtype=torch.int32, device=dev),
                occupied=torch.empty(N, dtype=torch.int32, device=dev), counts=torch.zeros(4, dtype=torch.int32, device=dev),
                occs=torch.empty(N, dtype=F32, device=dev), thr=torch.empty(2 + 2 * 256, dtype=F32, device=dev))
        bricks = ob["bricks"]
        _ops.grid_bricks(binary, out=bricks)  # the bitfield of the CURRENT grid (cached per tensor version)
        n_occ, n_cells = ob["counts"][0:1], ob["counts"][1:2]
        table = enc.table_half(enc.params)
        blob = self.bg_geo.build(requires_grad=False)
        sphere = ContractionType.UN_BOUNDED_SPHERE.value
        with device_guard(dev):
            s = stream_ptr()
            ob["jitter"][:cap * 3].uniform_()
            if not all_cells:
                if SORTED_REFRESH:
                    sorted_uniform_(ob["u"][:n_uniform])
                    sorted_uniform_(ob["u"][n_uniform:])
                else:
                    ob["u"].uniform_()
            check(lib.nsr_occupancy_select_cells(ptr(bricks), rx, ry, rz, ptr(ob["u"][:n_uniform]),
                                                 ptr(ob["u"][n_uniform:]), ptr(ob["jitter"]), n_uniform, int(all_cells) | 2,
                                                 cap, ptr(ob["brick_offset"]), ptr(ob["occupied"]), ptr(n_occ),
                                                 ptr(ob["cells"]), ptr(ob["x_unit"]), ptr(n_cells), s),
                  "nsr_occupancy_select_cells")
            # (flag 2: samples outside the unit sphere -- no world position -- are dropped by the selection)
            check(lib.nsr_contract_inv(ptr(ob["x_unit"]), ptr(grid.roi_aabb), sphere, ptr(ob["world"]), cap, s),
                  "nsr_contract_inv")
            check(lib.nsr_contract_to_unisphere(ptr(ob["world"]), self.radius, sphere, ptr(ob["x01"]), cap, s),
                  "nsr_contract_to_unisphere")
            check(lib.nsr_hashgrid_forward_ex(ptr(ob["x01"]), ptr(table), ptr(ob["enc"]), cap, 0, 2, desc.n_levels,
                                              _byref(desc), ptr(n_cells), s), "nsr_hashgrid_forward_ex")
            # the density head on the encoding alone (x == NULL), column 0 of every row
            check(lib.nsr_vmlp_forward(_byref(self.bg_geo.desc), ptr(blob), None, 0, ptr(ob["enc"]),
                                       0x40000000 | int(desc.n_features), None, ptr(ob["logit"]), None, cap, 0, ptr(n_cells), s),
                  "nsr_vmlp_forward(bg occupancy)")
            check(lib.nsr_occupancy_density_values_sphere(ptr(ob["logit"]), ptr(ob["x_unit"]), float(self.bg_bias),
                                                          float(m.render_step_size_bg), ptr(ob["occ"]), cap, ptr(n_cells), s),
                  "nsr_occupancy_density_values_sphere")
            check(lib.nsr_occupancy_update_values(ptr(ob["occ"]), float(ema_decay), float(occ_thre), ptr(ob["cells"]),
                                                  ptr(grid.occs), ptr(ob["occs"]), ptr(binary.view(torch.uint8)),
                                                  ptr(ob["thr"]), N, cap, ptr(n_cells), s), "nsr_occupancy_update_values")
            grid.occs.copy_(ob["occs"])
            check(lib.nsr_grid_pack_bricks(ptr(binary.view(torch.uint8)), rx, ry, rz, ptr(bricks), s),
                  "nsr_grid_pack_bricks")
        try:  # the cache of ops.grid_bricks keys on the tensor version, which a raw-pointer write does not bump
            binary._nsr_bricks = (binary._version, binary.data_ptr(), bricks)
        except Exception:  # noqa: BLE001
            pass

    @torch.no_grad()
    def surface_attributes(self, points):
        """``VolumeSDF.forward(points, with_grad=True, with_feature=True)`` + the "albedo" query of ``NeuSModel.export``
        (models/neus.py:313-323: colour network with viewing direction = -normal) on world points [n, 3], gradients off.
        -> dict(sdf [n], sdf_grad [n, 3], normal [n, 3], feature [n, feature_dim], rgb [n, 3])"""
        enc, desc, dev = self.enc, self.enc.grid_desc, points.device
        n = points.shape[0]
        T = 7 if self.fd else 1
        eps = self._fd_eps() if self.fd else 0.0
        mc = self._mask_count()
        with device_guard(dev):
            s = stream_ptr()
            pts = points.float().contiguous()
            zeros3, zeros1 = torch.zeros((n, 3), dtype=F32, device=dev), torch.zeros(n, dtype=F32, device=dev)
            ri = torch.arange(n, dtype=torch.int64, device=dev)
            x7 = torch.empty((T * n, 3), dtype=F32, device=dev)
            # positions (+ the six clamped taps) through the same kernel as the training step: origin = point, t = 0
            check(lib.nsr_neus_points(ptr(pts), ptr(zeros3), ptr(ri), ptr(zeros1), ptr(zeros1), self.radius, eps,
                                      int(self.fd), ptr(x7), None, n, None, s), "nsr_neus_points")
            table = enc.table_half(enc.params)
            encd = torch.empty((T * n, self.n_enc), dtype=F16, device=dev)
            sd = self.sdf.desc
            P = int(sd.in_pad)
            out = torch.empty((n, 16), dtype=F32, device=dev)
            blob = self.sdf.build(requires_grad=False)
            taps = g_in = dx01 = laplace = None
            if self.fd:
                check(lib.nsr_hashgrid_forward_taps(ptr(x7), ptr(table), ptr(encd), n, self.n_enc, 0, mc, _byref(desc), None,
                                                    s), "nsr_hashgrid_forward_taps")
                taps = torch.empty(6 * n, dtype=F32, device=dev)
                laplace = torch.empty(n, dtype=F32, device=dev)
            else:
                jac = torch.empty(n * self.n_enc * 3, dtype=F32, device=dev)
                check(lib.nsr_hashgrid_forward_jac(ptr(x7), ptr(table), ptr(encd), n, self.n_enc, 0, mc, _byref(desc),
                                                   ptr(jac), None, s), "nsr_hashgrid_forward_jac")
                g_in = torch.empty((n, P), dtype=F32, device=dev)
            check(lib.nsr_vmlp_forward(_byref(sd), ptr(blob), ptr(x7), 3, ptr(encd), self.n_enc, ptr(out), ptr(taps), ptr(g_in),
                                       T * n, n, None, s), "nsr_vmlp_forward(sdf)")
            if not self.fd:
                dx01 = torch.empty((n, 3), dtype=F32, device=dev)
                check(lib.nsr_hashgrid_jac_apply(ptr(jac), n, _byref(desc), _off(g_in, 3), P, ptr(dx01), None, None, 0, None,
                                                 s), "nsr_hashgrid_jac_apply(J^T dy)")
            acc = torch.zeros(16, dtype=F32, device=dev)
            inv_s = self._inv_s()
            grad, normal = torch.empty((n, 3), dtype=F32, device=dev), torch.empty((n, 3), dtype=F32, device=dev)
            alpha = torch.empty(n, dtype=F32, device=dev)
            tex_f32 = not self.tex_fused
            tex_in = torch.empty((n, 32), dtype=F32 if tex_f32 else F16, device=dev)
            dirs = zeros3
            for _ in range(2):  # pass 1: the normal; pass 2: the colour-network input with viewing direction = -normal
                check(lib.nsr_neus_shade_forward(ptr(out), ptr(g_in), P, ptr(dx01), ptr(taps), eps, self.radius, ptr(dirs),
                                                 ptr(zeros1), ptr(zeros1), ptr(inv_s), 1.0, self.n_feat, 1.0, ptr(grad),
                                                 ptr(normal), ptr(alpha), ptr(laplace), ptr(tex_in), int(tex_f32), ptr(acc),
                                                 n, None, s), "nsr_neus_shade_forward")
                dirs = (-normal).contiguous()
            if self.tex_fused:
                rgb_raw, _ = _ops.mlp_forward(tex_in, self.tex.half_params(self.tex.params), self.tex.mlp_desc, save_acts=False)
            else:
                rgb_raw = torch.empty((n, 16), dtype=F32, device=dev)
                tb = self.tex.build(requires_grad=False)
                check(lib.nsr_vmlp_forward(_byref(self.tex.desc), ptr(tb), ptr(tex_in), 32, None, 0, ptr(rgb_raw), None, None,
                                           n, n, None, s), "nsr_vmlp_forward(texture)")
            rgb = torch.sigmoid(rgb_raw[:, :3].float())
        return {"sdf": out[:, 0], "sdf_grad": grad, "normal": normal, "feature": out[:, :self.n_feat], "rgb": rgb}

    def forward_backward(self, rays, gt_rgb, fg_mask, background, compute_grads=True, loss_scale=1.0, march_handle=None,
                         after_march=None):
        """forward + the system's loss terms (systems/neus.py:96-130, weights = ``loss_weights``) + backward in one go.
        Gradient contract: ``.grad`` of the hash tables is OVERWRITTEN (the owner-computes backward writes every entry),
        ``.grad`` of every other parameter (fp32 heads through the weight-norm fold, variance, fused colour MLP) is created
        when it is None and ADDED to otherwise -- call with ``.grad = None`` (or zeroed) parameters, as the trainers do."""
        gen = self._step(rays, gt_rgb, fg_mask, background, compute_grads, loss_scale, march_handle, after_march, False)
        try:
            next(gen)
        except StopIteration as stop:  # no gradients asked for / nothing to differentiate
            return stop.value
        try:
            gen.send(None)
        except StopIteration as stop:
            return stop.value
        raise RuntimeError("FusedNeuSStep: the step generator did not finish")

    def release_gradient_buffers(self):
        """the fp32 heads' ``.grad`` tensors are views of one persistent buffer per network (VanillaBlob.push_gradient): a
        caller that hands those tensors on (autograd) takes the buffers with them, the next step allocates new ones"""
        for vb in (self.sdf, None if self.tex_fused else self.tex, getattr(self, "bg_geo", None), getattr(self, "bg_tex", None)):
            if vb is not None:
                vb._grad_flat = None

    def render(self, rays, background, need_grad, march_handle=None):
        """the step split AT THE LOSS (nsr.models.FusedNeuSModel: the reference's system owns loss and backward()).
        -> (res, finish): ``res`` the forward outputs; ``finish(upstream)`` (None when ``need_grad`` is false or nothing was
        marched) runs the backward from a dict of upstream gradients -- comp_rgb_full / comp_rgb [R,3], opacity / depth [R,1],
        weights / sdf_samples / sdf_laplace_samples [N], sdf_grad_samples [N,3] (missing = zero) -- and leaves the parameter
        gradients in ``.grad`` like ``forward_backward`` does."""
        gen = self._step(rays, None, None, background, need_grad, 1.0, march_handle, None, True)
        try:
            res = next(gen)
        except StopIteration as stop:
            return stop.value, None

        def finish(upstream):
            try:
                gen.send(upstream or {})
            except StopIteration:
                return
            raise RuntimeError("FusedNeuSStep: the step generator did not finish")

        return res, finish

    def _step(self, rays, gt_rgb, fg_mask, background, compute_grads, loss_scale, march_handle, after_march, external):
        """generator: runs the forward, yields the result dict, is sent the upstream gradients (``external``) or None (built-in
        loss terms) and runs the backward.  No torch context manager is held across the yield."""
        self.adam_applied, self.bf16_written, self.grad_written = set(), set(), set()
        m, enc, lw = self.model, self.enc, self.loss_weights
        dev = rays.device
        n_rays = rays.shape[0]
        grid = m.occupancy_grid
        desc = enc.grid_desc
        with torch.no_grad(), device_guard(dev):
            s = stream_ptr()
            step_start = torch.cuda.Event()
            step_start.record(torch.cuda.current_stream())
            if march_handle is None:
                rays_o, rays_d = rays[:, 0:3].contiguous(), rays[:, 3:6].contiguous()
                march_handle = self.march_begin(rays_o, rays_d)
                if self.bg:
                    march_handle = (march_handle, self.bg_march_begin(rays_o, rays_d))
            bg_handle = None
            if self.bg:
                march_handle, bg_handle = march_handle
            rays_o, rays_d = march_handle.args[0], march_handle.args[1]
            packed, ri, t0, t1 = _ops.ray_march_finish(march_handle)
            N = ri.shape[0]
            bgc = None
            self._n_samples = N
            T = 7 if self.fd else 1
            eps = self._fd_eps() if self.fd else 0.0
            mc = self._mask_count()
            x7 = torch.empty((T * N, 3), dtype=F32, device=dev)
            dirs = torch.empty((N, 3), dtype=F32, device=dev)
            check(lib.nsr_neus_points(ptr(rays_o), ptr(rays_d), ptr(ri), ptr(t0), ptr(t1), self.radius, eps, int(self.fd),
                                      ptr(x7), ptr(dirs), N, None, s), "nsr_neus_points")
            table = enc.table_half(enc.params)
            positions_ready = torch.cuda.Event()
            positions_ready.record(torch.cuda.current_stream())
            # encoding of the T N points, TILE-major [T N / 16][L][16][F] (masked levels: zeros): the 16 rows of an MFMA tile
            # are one contiguous KB as in a row-major [T N][C] array, but inside it every level's 16 x F halfs are contiguous,
            # so the encode kernels (lane = (point, level)) store 64-B runs instead of 4-B pieces 64 B apart -- row-major
            # costs ten times the output in fabric writes -- and the fp32 MLP kernels read 2-3 runs per load instead of 16
            # rows.  (Level-major [L][T N][F] had the encode's gain, 36 us plain / 131 us taps, but cost the MFMA kernels
            # 180 / 660 us: a tile's operands then sit in 16 far-apart lines.)  NSR_NEUS_ENC_LAYOUT=0 selects row-major.
            lay = ENC_LAYOUT
            encd = torch.empty(((T * N + 15) // 16 * 16, self.n_enc), dtype=F16, device=dev)
            tws = None
            if self.fd and self.fold_taps and compute_grads and N > 0:
                # stencil mode of the table backward: the 7-tap encode leaves the taps' crossing masks in its tap workspace
                tws = torch.empty(int(lib.nsr_hashgrid_backward_params_taps_workspace_floats(_byref(desc), N)), dtype=F32,
                                  device=dev)
                check(lib.nsr_hashgrid_forward_taps_masks(ptr(x7), ptr(table), ptr(encd), N, self.n_enc, lay, mc,
                                                          _byref(desc), ptr(tws), s), "nsr_hashgrid_forward_taps_masks")
                positions_ready = torch.cuda.Event()  # (the binning on the helper stream now waits for the encode's masks)
                positions_ready.record(torch.cuda.current_stream())
            elif self.fd:  # the sample's corners are gathered once and shared with its six taps
                check(lib.nsr_hashgrid_forward_taps(ptr(x7), ptr(table), ptr(encd), N, self.n_enc, lay, mc, _byref(desc),
                                                    None, s), "nsr_hashgrid_forward_taps")
            else:  # analytic normals: keep the per-level Jacobian (384 B / sample) instead of two more table gathers
                jac = torch.empty(N * self.n_enc * 3, dtype=F32, device=dev)
                check(lib.nsr_hashgrid_forward_jac(ptr(x7), ptr(table), ptr(encd), N, self.n_enc, lay, mc, _byref(desc),
                                                   ptr(jac), None, s), "nsr_hashgrid_forward_jac")
            gws = bin_event = None
            if compute_grads and N > 0:
                # the table backward's binning needs only the positions: it runs on a helper stream underneath the whole
                # forward pass (count / scan / fill: 0.18 ms at 5e5 points, 0.5 ms at the 7 N points of the C5 stencil);
                # queued AFTER the encode so that its host-side setup does not hold the main stream's first big kernel back
                nws = int(lib.nsr_hashgrid_backward_params_workspace_floats(_byref(desc), T * N))
                gws = torch.empty(nws, dtype=F32, device=dev)
                if getattr(self, "_helper", None) is None:
                    self._helper = _shared_stream(dev, "helper")
                self._helper.wait_event(positions_ready)
                with torch.cuda.stream(self._helper):
                    if tws is not None:  # in-cell taps are folded into their sample's items (stencil mode)
                        check(lib.nsr_hashgrid_backward_params_owner_bin_taps_masked(ptr(x7), ptr(gws), ptr(tws), N, mc,
                                                                                     _byref(desc), stream_ptr()),
                              "nsr_hashgrid_backward_params_owner_bin_taps_masked")
                        tws.record_stream(self._helper)
                    else:
                        bin_fn = lib.nsr_hashgrid_backward_params_owner_bin if self.fd else \
                            lib.nsr_hashgrid_backward_params_owner_bin_second_order  # (same slice configuration as the pass)
                        check(bin_fn(ptr(x7), ptr(gws), T * N, mc, _byref(desc), None, stream_ptr()),
                              "nsr_hashgrid_backward_params_owner_bin")
                    bin_event = torch.cuda.Event()
                    bin_event.record(self._helper)
                gws.record_stream(self._helper)
                x7.record_stream(self._helper)
        sdf_blob = self.sdf.build(requires_grad=compute_grads)
        tex_blob = None if self.tex_fused else self.tex.build(requires_grad=compute_grads)
        if self.bg:
            self._bg_blob_t = self.bg_tex.build(requires_grad=compute_grads)
        with torch.no_grad(), device_guard(dev):
            out = torch.empty((N, 16), dtype=F32, device=dev)
            taps = torch.empty(6 * N, dtype=F32, device=dev) if self.fd else None
            sd = self.sdf.desc
            P, C, F = int(sd.in_pad), self.n_enc, int(desc.n_features)
            # how the MLP kernels read the encoding: a row stride, or 0x40000000 | F = tile-major (0x80000000 | F: level-major)
            ENC_LM = (0x40000000 | F) if lay == 2 else C
            g_in = None if self.fd else torch.empty((N, P), dtype=F32, device=dev)
            check(lib.nsr_vmlp_forward(_byref(sd), ptr(sdf_blob.detach()), ptr(x7), 3, ptr(encd), ENC_LM, ptr(out), ptr(taps),
                                       ptr(g_in), T * N, N, None, s), "nsr_vmlp_forward(sdf)")
            # the background's pruning pass decides its sample count (num_samples_full = N + S).  It is queued HERE, on the
            # background stream, once the foreground's positions / encode / SDF network are in the main stream's queue: queued
            # first, its dozen launches kept the main stream idle for as long as the host needed for them.
            # A trainer's own step (defer_bg_count): the kept count stays on the device (_bg_prune_defer); `hook` below queues
            # the next batch once the count has reached the host -- polled once, where the main stream is fullest, else waited for by
            # the trainer when the whole step is queued.
            defer, hook = False, None
            if self.bg:
                self._bg_grads = compute_grads
                self._bg_uses(rays_o, rays_d)
                with self._bg_ctx(dev, behind=step_start):  # (behind the previous step's optimizer and a grid refresh only)
                    bgc = self._bg_prune_begin(bg_handle)
                    defer = self.defer_bg_count and not external and bgc["M"] > 0
                    if defer:
                        self._bg_prune_defer(bgc)
                    else:
                        self._bg_prune_finish(bgc)
            if defer and after_march is not None:
                def hook(block=False):
                    if "_count" not in bgc or not (block or _ops.read_count_ready(bgc["_count"])):
                        return
                    bgc["S"] = _ops.read_count_finish(bgc.pop("_count"))
                    after_march(N + bgc["S"])
            if after_march is not None and not defer:
                # the sample count of this step is known: a trainer queues the next batch's ray preparation + marching
                # (side stream) here -- AFTER the first ~0.6 ms of this step's kernels are in the queue, so that the host
                # time it takes does not leave the main stream idle.  (Later is worse: measured, the marching pass then
                # runs next to the backward kernels, and the table backward is sensitive to co-runners.)
                after_march(N + (bgc["S"] if bgc else 0))
            dx01 = None
            if not self.fd:  # J^T (d sdf / d encoding): models/geometry.py:176-180 through the encoder
                dx01 = torch.empty((N, 3), dtype=F32, device=dev)
                # (with gradients: the kernel also leaves d sdf / d encoding level-major -- its tile is in LDS anyway -- which
                # is how the table backward's second-order term reads it: no transposing pass there)
                g_lm = torch.empty(self.n_enc * N, dtype=F32, device=dev) if compute_grads else None
                check(lib.nsr_hashgrid_jac_apply_ex(ptr(jac), N, _byref(desc), _off(g_in, 3), P, ptr(dx01), None, None, 0,
                                                    ptr(g_lm), None, s), "nsr_hashgrid_jac_apply(J^T dy)")
            acc = torch.zeros(16, dtype=F32, device=dev)
            inv_s = self._inv_s()
            anneal = float(getattr(m, "cos_anneal_ratio", 1.0))
            grad = torch.empty((N, 3), dtype=F32, device=dev)
            normal = torch.empty((N, 3), dtype=F32, device=dev)
            alpha = torch.empty(N, dtype=F32, device=dev)
            laplace = torch.empty(N, dtype=F32, device=dev) if self.fd else None
            tex_f32 = not self.tex_fused
            tex_in = torch.empty((N, 32), dtype=F32 if tex_f32 else F16, device=dev)
            check(lib.nsr_neus_shade_forward(ptr(out), ptr(g_in), P, ptr(dx01), ptr(taps), eps, self.radius, ptr(dirs),
                                             ptr(t0), ptr(t1), ptr(inv_s), anneal, self.n_feat,
                                             float(lw.get("sparsity_scale", 1.0)), ptr(grad), ptr(normal), ptr(alpha),
                                             ptr(laplace), ptr(tex_in), int(tex_f32), ptr(acc), N, None, s),
                  "nsr_neus_shade_forward")
            # colour network
            if self.tex_fused:
                tex = self.tex
                w2 = tex.half_params(tex.params)
                rgb_raw, acts2 = _ops.mlp_forward(tex_in, w2, tex.mlp_desc, save_acts=compute_grads)
            else:
                td = self.tex.desc
                rgb_raw = torch.empty((N, 16), dtype=F32, device=dev)
                check(lib.nsr_vmlp_forward(_byref(td), ptr(tex_blob.detach()), ptr(tex_in), 32, None, 0, ptr(rgb_raw), None,
                                           None, N, N, None, s), "nsr_vmlp_forward(texture)")
            bg = background.to(F32).contiguous()
            bg_arg, bg_stride, op_bg = bg, 0, None
            if bgc is not None:  # per-ray background = the NeRF++ branch's colour (models/neus.py:273-283)
                self._bg_uses(bg)
                with self._bg_ctx(dev):  # (behind the main stream: `bg` above, and a model-entry caller's own work)
                    self._bg_forward(bgc, bg)
                self._bg_join([v for v in bgc.values()])
                bg_arg, bg_stride, op_bg = bgc["comp_rgb"], 3, bgc["opacity"]
            n1 = max(N, 1)  # the ray kernels run for every ray: no NULL sample arrays when nothing was marched
            weights, trans = torch.empty(n1, dtype=F32, device=dev), torch.empty(n1, dtype=F32, device=dev)
            comp_rgb = torch.empty((n_rays, 3), dtype=F32, device=dev)
            comp_normal = torch.empty((n_rays, 3), dtype=F32, device=dev)
            comp_full = torch.empty((n_rays, 3), dtype=F32, device=dev)
            opacity = torch.empty((n_rays, 1), dtype=F32, device=dev)
            depth = torch.empty((n_rays, 1), dtype=F32, device=dev)
            check(lib.nsr_neus_composite_forward(ptr(packed), ptr(alpha), ptr(rgb_raw), int(tex_f32), ptr(normal), ptr(t0),
                                                 ptr(t1), ptr(bg_arg), bg_stride, ptr(weights), ptr(trans), ptr(comp_rgb),
                                                 ptr(opacity), ptr(depth), ptr(comp_normal), ptr(comp_full), n_rays, s),
                  "nsr_neus_composite_forward")
            gt = None if gt_rgb is None else gt_rgb.to(F32).contiguous()
            fg = None if fg_mask is None else fg_mask.to(F32).contiguous()
            if gt is not None:  # (a caller-owned loss needs no per-ray loss sums)
                check(lib.nsr_neus_loss_rays(ptr(comp_full), ptr(opacity), ptr(op_bg), ptr(gt), ptr(fg), ptr(acc), n_rays, None,
                                             s), "nsr_neus_loss_rays")
            lean = getattr(self, "lean_outputs", False) and compute_grads  # a trainer does not read the validity masks
            res = {"comp_rgb": comp_rgb, "comp_normal": comp_normal, "opacity": opacity, "depth": depth,
                   "rays_valid": None if lean else opacity > 0, "comp_rgb_full": comp_full,
                   "rays_valid_full": None if lean else opacity > 0,
                   "num_samples": N, "sdf_samples": out[:, 0], "sdf_grad_samples": grad, "weights": weights[:N],
                   "ray_indices": ri, "t_starts": t0, "t_ends": t1, "alpha": alpha, "loss_acc": acc,
                   "inv_s": inv_s[0]}
            if self.fd:
                res["sdf_laplace_samples"] = laplace
            if bgc is not None:
                res.update({"comp_rgb_bg": bgc["comp_rgb"], "opacity_bg": bgc["opacity"], "depth_bg": bgc["depth"],
                            "rays_valid_bg": None if lean else bgc["opacity"] > 0, "num_samples_bg": bgc["S"],
                            "num_marched_bg": bgc["M"], "num_samples_full": None if defer else N + bgc["S"],
                            "rays_valid_full": None if lean else (opacity > 0) | (bgc["opacity"] > 0),
                            "weights_bg": bgc["weights"], "ray_indices_bg": bgc["ri"], "t_starts_bg": bgc["t0"],
                            "t_ends_bg": bgc["t1"]})
                if defer:
                    res["_bg_deferred"] = (bgc, hook)
            done = not compute_grads or (N == 0 and bgc is None)
        if done:
            return res
        # ---- the loss boundary: nothing of torch's thread state (no_grad, current device) is held across this yield ----------
        upstream = yield res
        ups = None
        if external:
            import nsr_hip
            ups, keep_up = nsr_hip.NsrNeusUpstream(), []
            for key, field in (("comp_rgb_full", "comp_rgb_full"), ("comp_rgb", "comp_rgb"), ("opacity", "opacity"),
                               ("depth", "depth"), ("weights", "weights"), ("sdf_samples", "sdf_samples"),
                               ("sdf_grad_samples", "sdf_grad_samples"), ("sdf_laplace_samples", "sdf_laplace_samples")):
                g_up = (upstream or {}).get(key)
                if g_up is not None:
                    g_up = g_up.detach().to(F32).contiguous()
                    keep_up.append(g_up)
                    setattr(ups, field, g_up.data_ptr())
        with torch.no_grad(), device_guard(dev):
            s = stream_ptr()
            # the built-in loss terms are off when the caller owns the loss
            lw8c = (ctypes.c_float * 8)(*[0.0 if (external and k != "sparsity_scale") else float(lw.get(k, 0.0))
                                          for k in LOSS_KEYS])
            upp = _byref(ups) if ups is not None else None
            d_alpha = torch.empty(n1, dtype=F32, device=dev)
            d_rgb = torch.empty((n1, 16), dtype=F32, device=dev)
            d_bg = None if bgc is None else torch.empty((n_rays, 3), dtype=F32, device=dev)
            check(lib.nsr_neus_composite_backward_ex(ptr(packed), ptr(alpha), ptr(rgb_raw), int(tex_f32), ptr(weights),
                                                     ptr(trans), ptr(bg_arg), bg_stride, ptr(op_bg), ptr(comp_full),
                                                     ptr(opacity), ptr(gt), ptr(fg), ptr(acc), lw8c, float(loss_scale),
                                                     ptr(d_alpha), ptr(d_rgb), ptr(d_bg), n_rays, None, upp, ptr(t0), ptr(t1),
                                                     s), "nsr_neus_composite_backward")
            g_bg = None
            if bgc is not None:  # beside the foreground's backward; joined before the gradients are pushed below
                self._bg_uses(d_bg)
                with self._bg_ctx(dev):
                    g_bg = self._bg_backward(bgc, d_bg)
            if N == 0:  # background only: nothing flows into the foreground networks
                self._bg_join(g_bg)
                return self._finish_bg_only(res, g_bg)
            # colour network backward -> d tex_in (fp32 [N, 32])
            if self.tex_fused:
                tex = self.tex
                if tex.params.grad is None:
                    tex.params.grad = torch.zeros_like(tex.params)
                d_tex = _ops.mlp_backward(d_rgb, rgb_raw, tex_in, acts2, w2, tex.mlp_desc, grad_weights=tex.params.grad,
                                          want_dx=True, grad_scale=self.grad_scale)
            else:
                td = self.tex.desc
                d_tex = torch.empty((N, 32), dtype=F32, device=dev)
                g_tex = torch.empty(self.tex.n_floats, dtype=F32, device=dev)
                ws = torch.empty(int(lib.nsr_vmlp_backward_workspace_floats(_byref(td), N)), dtype=F32, device=dev)
                check(lib.nsr_vmlp_backward(_byref(td), ptr(tex_blob.detach()), ptr(tex_in), 32, None, 0, ptr(d_rgb), None,
                                            None, ptr(d_tex), 32, 0, 32, 0, ptr(g_tex), 0, ptr(ws), N, N, None, s),
                      "nsr_vmlp_backward(texture)")
            d_out = torch.empty((N, 16), dtype=F32, device=dev)
            gx = p_in = d_taps = None
            if self.fd:
                d_taps = torch.empty(6 * N, dtype=F32, device=dev)
            else:
                gx = torch.empty((N, 3), dtype=F32, device=dev)
                p_in = torch.zeros((N, P), dtype=F32, device=dev)
            check(lib.nsr_neus_shade_backward_ex(ptr(out), ptr(grad), ptr(normal), ptr(dirs), ptr(t0), ptr(t1), ptr(inv_s),
                                                 anneal, ptr(laplace), eps, self.radius, ptr(d_alpha), ptr(d_tex),
                                                 self.n_feat, lw8c, float(loss_scale), float(N), ptr(d_out), ptr(gx),
                                                 ptr(p_in), P, ptr(d_taps), ptr(acc), N, None, upp, s),
                  "nsr_neus_shade_backward")
            if enc.params.grad is None:
                enc.params.grad = torch.zeros_like(enc.params)
            g_table = enc.params.grad
            # analytic normals: the encoder's input gradient is differentiated again -- d_dy = J gx joins p_in (what flows on
            # into the SDF network), the second-order table gradient is added after the first-order one below
            # SDF network backward: d enc (level-major, what the owner-computes table backward reads), dW
            d_enc = torch.empty(C * T * N, dtype=F32, device=dev)
            g_sdf = torch.empty(self.sdf.n_floats, dtype=F32, device=dev)
            ws = torch.empty(int(lib.nsr_vmlp_backward_workspace_floats(_byref(sd), T * N)), dtype=F32, device=dev)
            if not self.fd:
                check(lib.nsr_hashgrid_jac_apply(ptr(jac), N, _byref(desc), None, 0, None, ptr(gx), _off(p_in, 3), P, None,
                                                 s), "nsr_hashgrid_jac_apply(J g)")
            check(lib.nsr_vmlp_backward(_byref(sd), ptr(sdf_blob.detach()), ptr(x7), 3, ptr(encd), ENC_LM, ptr(d_out),
                                        ptr(d_taps), ptr(p_in), ptr(d_enc), 0, 3, C, F, ptr(g_sdf), 0, ptr(ws), T * N, N,
                                        None, s), "nsr_vmlp_backward(sdf)")
            if hook is not None:
                # the one place where the step polls for the background's kept count: the main stream holds ~0.5 ms of queued
                # kernels here (colour backward, shade backward, SDF backward), more than queueing the next batch costs the host
                hook()
            torch.cuda.current_stream().wait_event(bin_event)  # the items are binned (helper stream)
            # a trainer on one GPU hands over AdamW for the table (self.table_adam): the owner workgroups apply it in their
            # write-out -- no 50 MB gradient store, no optimizer sweep over the table
            ad = (self.table_adam or {}).get("fg")
            bf = None if ad is not None else (self.table_bf16 or {}).get("fg")
            if bf is not None:
                if self.fd and tws is not None:
                    check(lib.nsr_hashgrid_backward_params_owner_accumulate_taps_bf16(
                        ptr(x7), ptr(d_enc), ptr(bf), ptr(gws), ptr(tws), N, mc, _byref(desc), s),
                        "nsr_hashgrid_backward_params_owner_accumulate_taps_bf16")
                elif self.fd:
                    check(lib.nsr_hashgrid_backward_params_owner_accumulate_range(
                        ptr(x7), ptr(d_enc), None, ptr(bf), ptr(gws), T * N, mc, 1.0, 0, desc.n_levels, _byref(desc), None, s),
                        "nsr_hashgrid_backward_params_owner_accumulate_range")
                else:
                    check(lib.nsr_hashgrid_backward_params_owner_with_second_order_bf16(
                        ptr(x7), ptr(d_enc), ptr(g_lm), 0, ptr(gx), ptr(bf), ptr(gws), N, mc, 1, _byref(desc), s),
                        "nsr_hashgrid_backward_params_owner_with_second_order_bf16")
                self.bf16_written.add("fg")
            elif self.fd and tws is not None:
                if ad is not None:
                    check(lib.nsr_hashgrid_backward_params_owner_accumulate_taps_adam(
                        ptr(x7), ptr(d_enc), ptr(gws), ptr(tws), N, mc, _byref(desc), _byref(ad), s),
                        "nsr_hashgrid_backward_params_owner_accumulate_taps_adam")
                else:
                    check(lib.nsr_hashgrid_backward_params_owner_accumulate_taps(ptr(x7), ptr(d_enc), ptr(g_table), ptr(gws),
                                                                                 ptr(tws), N, mc, 0, _byref(desc), s),
                          "nsr_hashgrid_backward_params_owner_accumulate_taps")
            elif self.fd:
                if ad is not None:
                    check(lib.nsr_hashgrid_backward_params_owner_accumulate_adam(ptr(x7), ptr(d_enc), 2, 0, ptr(gws), T * N, mc,
                                                                                 1.0, _byref(desc), None, _byref(ad), s),
                          "nsr_hashgrid_backward_params_owner_accumulate_adam")
                else:
                    check(lib.nsr_hashgrid_backward_params_owner_accumulate(ptr(x7), ptr(d_enc), 2, 0, ptr(g_table), ptr(gws),
                                                                            T * N, mc, 1.0, 0, _byref(desc), None, s),
                          "nsr_hashgrid_backward_params_owner_accumulate")
            elif ad is not None:
                check(lib.nsr_hashgrid_backward_params_owner_with_second_order_adam(
                    ptr(x7), ptr(d_enc), ptr(g_lm), 0, ptr(gx), ptr(gws), N, mc, 1, _byref(desc), _byref(ad), s),
                    "nsr_hashgrid_backward_params_owner_with_second_order_adam")
            else:  # first- and second-order table gradients share their items: one accumulation pass
                check(lib.nsr_hashgrid_backward_params_owner_with_second_order(
                    ptr(x7), ptr(d_enc), ptr(g_lm), 0, ptr(gx), ptr(g_table), ptr(gws), N, mc, 0, 1, _byref(desc), s),
                    "nsr_hashgrid_backward_params_owner_with_second_order")
            if ad is not None:
                self.adam_applied.add("fg")
            elif bf is None:
                self.grad_written.add("fg")
        # weight norm / bias gradients through the host-side fold
        if g_bg is not None:
            self._bg_join(g_bg)
            self.bg_geo.push_gradient(g_bg[0])
            self.bg_tex.push_gradient(g_bg[1])
        self.sdf.push_gradient(g_sdf)
        if not self.tex_fused:
            self.tex.push_gradient(g_tex)
        var = m.variance.variance
        # inv_s = exp(10 v) (models/neus.py:27-32); the clip(1e-6, 1e6) is handled in the kernel
        fresh = var.grad is None
        if fresh:
            var.grad = torch.empty_like(var)
        with device_guard(dev):
            check(lib.nsr_neus_variance_gradient(ptr(acc), ptr(inv_s), ptr(var.grad), 0 if fresh else 1, stream_ptr()),
                  "nsr_neus_variance_gradient")
        return res


def neus_lr_scale(step, config_name, max_steps=20000):
    """learning-rate factor of the reference's SequentialLR (interval: step) at optimizer step ``step`` (0-based):
    neus-blender.yaml: LinearLR 0.01 -> 1 over 500 steps, then ExponentialLR 0.1^(1 / (max_steps - 500));
    neus-dtu.yaml / neuralangelo-*.yaml: constant for 5000 steps, then ExponentialLR 0.1^(1 / (max_steps - 5000))"""
    if config_name in ("neuralangelo", "neus-dtu"):
        c = 5000
        return 1.0 if step < c else 0.1 ** ((step - c) / (max_steps - c))
    w = 500
    if step < w:
        return 0.01 + (1.0 - 0.01) * step / w
    return 0.1 ** ((step - w) / (max_steps - w))


class SmallAdamW:
    """torch.optim.AdamW (lr per tensor, betas (0.9, 0.99), eps 1e-15, weight decay 0.01: systems/utils.py:314-325 with
    the YAML's optimizer.params groups) over the handful of small fp32 tensors of a NeuS model -- fp32 heads, variance --
    as ONE launch (nsr_adamw_multi) instead of the ~16 foreach launches + their host time.  Tensors whose ``.grad`` is
    None are skipped, like torch does; ``step`` leaves the gradients set to None."""

    def __init__(self, params_and_lrs, betas=(0.9, 0.99), eps=1e-15, weight_decay=0.01):
        self.items = [(p, float(lr)) for p, lr in params_and_lrs if p.numel() > 0]
        if len(self.items) > 32:
            raise NotImplementedError("SmallAdamW: at most 32 tensors")
        self.betas, self.eps, self.wd = betas, eps, weight_decay
        self.step_count = 0
        # ``device_step``: the step count (and the bias corrections) live on the device, where an overflow guard can hold them
        # back for a skipped step (nsr_adamw_multi); ``step_count`` then counts the calls, skipped ones included
        self._dev_state = None
        n = sum(p.numel() for p, _ in self.items)
        dev = self.items[0][0].device if self.items else None
        self._m = torch.zeros(n, dtype=F32, device=dev) if self.items else None
        self._v = torch.zeros(n, dtype=F32, device=dev) if self.items else None
        self.state, off = {}, 0
        for p, _ in self.items:
            if p.dtype != F32 or not p.is_contiguous():
                raise NotImplementedError("SmallAdamW: contiguous fp32 parameters")
            self.state[p] = (self._m[off:off + p.numel()], self._v[off:off + p.numel()])
            off += p.numel()

    def _device_state(self):
        if self._dev_state is None:
            dev = self.items[0][0].device
            self._dev_state = (torch.full((4,), self.step_count, dtype=torch.int32, device=dev)[:1],
                               torch.zeros(4, dtype=F32, device=dev))
        return self._dev_state

    def taken_steps(self):
        """optimizer steps really taken (a device-side count excludes the ones an overflow guard skipped) -- synchronises"""
        return self.step_count if self._dev_state is None else int(self._dev_state[0].item())

    def step(self, lr_scale=1.0, device_step=False):
        live = [(p, lr) for p, lr in self.items if p.grad is not None]
        if device_step or self._dev_state is not None:
            step_dev, hyper_dev = self._device_state()
        else:
            step_dev = hyper_dev = None
        self.step_count += 1
        if not live and step_dev is None:
            return
        segs = (NsrAdamSegment * max(len(live), 1))()
        for sg, (p, lr) in zip(segs, live):
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            m, v = self.state[p]
            sg.params, sg.grad, sg.exp_avg, sg.exp_avg_sq = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
            sg.n, sg.lr = p.numel(), lr * lr_scale
        bc1, bc2 = 1.0 - self.betas[0] ** self.step_count, 1.0 - self.betas[1] ** self.step_count
        with device_guard(self.items[0][0].device):
            check(lib.nsr_adamw_multi(segs, len(live), self.betas[0], self.betas[1], self.eps, self.wd, bc1, bc2, 0,
                                      ptr(step_dev), ptr(hyper_dev), stream_ptr()), "nsr_adamw_multi")
        for p, _ in live:
            p.grad = None

    def state_dict(self):
        return {"step_count": self.taken_steps(), "m": None if self._m is None else self._m.detach().clone(),
                "v": None if self._v is None else self._v.detach().clone()}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step_count"])
        if self._dev_state is not None:
            self._dev_state[0].fill_(self.step_count)
        if self._m is not None and sd.get("m") is not None:
            self._m.copy_(sd["m"])
            self._v.copy_(sd["v"])


class NeuSTrainer:
    """one-process-per-GPU training step of the reference's NeuSSystem (systems/neus.py:87-152) on the fused runner:
    sample rays -> schedules / occupancy refresh -> fused forward + losses + backward -> (grad all-reduce) -> AdamW with the
    per-group learning rates of the YAML (geometry / texture 0.01, variance 0.001).  The marching pass of step t+1 runs
    on a side stream underneath step t (it needs rays and the occupancy grid only), except across a grid refresh."""

    def __init__(self, model, dataset, config, loss_weights, config_name="neus-blender", rank=0, world_size=1, seed=42,
                 max_steps=20000):
        import tinycudann as tcnn
        from .parallel import broadcast_parameters, shard_seed
        from .trainer import FusedAdamW
        self.model, self.dataset, self.config, self.config_name = model, dataset, config, config_name
        self.rank, self.world_size, self.max_steps = rank, world_size, max_steps
        self.device = next(model.parameters()).device
        self.fused = FusedNeuSStep(model, loss_weights)
        # systems/neus.py:28: the dynamic ray count targets rays x (foreground + background samples per ray)
        self.train_num_samples = config["train_num_rays"] * (config["num_samples_per_ray"] +
                                                             config.get("num_samples_per_ray_bg", 0))
        self.train_num_rays = config["train_num_rays"]
        self.global_step = 0
        self.seed = int(seed)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(shard_seed(seed, rank))
        if world_size > 1:
            broadcast_parameters(model)
        tc = [m for m in model.modules() if isinstance(m, tcnn.Module) and m.params.numel() > 0]
        tc_ids = {id(m.params) for m in tc}
        var = [model.variance.variance]
        rest = [p for p in model.parameters() if id(p) not in tc_ids and p is not var[0] and p.numel() > 0]
        self.opt = FusedAdamW(tc, [], lr=0.01)
        # multi-GPU: the hash tables' gradients are reduce-scattered in bf16, each rank steps its shard and the fp16
        # images are all-gathered (nsr.parallel.ShardedAdamW); the small fp32 heads + variance are all-reduced
        self.sharded, self.comm_timings = None, None
        if world_size > 1 or os.environ.get("NSR_FORCE_SHARDED"):  # (the switch: see nsr/trainer.py)
            import torch.distributed as dist
            from .parallel import ShardedAdamW
            if dist.is_initialized():
                from .trainer import guard_stale_state_dict
                # NSR_TRANSPORT=fp32: the table gradients travel as fp32 (the A/B of the bf16 wire format, as in nsr/trainer.py)
                fp32 = os.environ.get("NSR_TRANSPORT", "bf16") == "fp32"
                self.sharded = ShardedAdamW(tc, lr=0.01, transport=torch.float32 if fp32 else torch.bfloat16)
                guard_stale_state_dict(model, self.sharded)
        self._tables = tuple(m for m in tc if getattr(m, "grid_desc", None) is not None)  # their backward OVERWRITES .grad
        # one GPU: AdamW on the tables runs inside their backward (NSR_NEUS_SEPARATE_ADAM=1: the stand-alone sweep, for A/B)
        self.fuse_table_adam = not os.environ.get("NSR_NEUS_SEPARATE_ADAM")
        self._table_of = {"fg": self.fused.enc}
        if self.fused.bg:
            self._table_of["bg"] = self.fused.bg_enc
        self._rest = rest + var
        self.opt_rest = SmallAdamW([(p, 0.01) for p in rest] + [(p, 0.001) for p in var])
        self.fused.lean_outputs = True  # no per-ray validity masks etc. in the step's result dict
        self.device_occupancy_refresh = not os.environ.get("NSR_NEUS_TORCH_REFRESH")  # foreground grid (A/B switch)
        self._pending, self._side, self._grids_ready = None, None, None
        self.fused.defer_bg_count = self.fused.bg and not os.environ.get("NSR_NEUS_BG_SYNC_COUNT") and \
            not os.environ.get("NSR_NEUS_BG_ON_MAIN")
        self.last = {}
        from .trainer import resync_after_model_load
        resync_after_model_load(self)

    def state_dict(self):
        """``model.state_dict()`` with every fp32 parameter current; at world > 1 a COLLECTIVE (every rank calls it): the
        tables' master values are gathered from the owners' optimizer shards first (nsr.parallel.ShardedAdamW)"""
        from .trainer import checkpoint_state_dict
        return checkpoint_state_dict(self.model, self.sharded)

    def save(self, path):
        """model + training state (optimizer moments of the tables and of the small fp32 tensors, step counts, dynamic ray
        count, sampler state); a COLLECTIVE at world > 1"""
        from .trainer import training_state
        sd, ts = self.state_dict(), training_state(self, (self.opt_rest,))
        if self.rank == 0:
            torch.save({"state_dict": sd, "global_step": self.global_step, "training_state": ts}, path)

    def load(self, path_or_ckpt):
        """resume (reference launch.py:112-113): weights, optimizer state, step counter.  Every rank calls it."""
        from .trainer import restore_training_state
        ck = torch.load(path_or_ckpt, map_location=self.device) if isinstance(path_or_ckpt, (str, bytes, os.PathLike)) \
            else path_or_ckpt
        self.model.load_state_dict(ck["state_dict"])
        if ck.get("training_state") is not None:
            restore_training_state(self, ck["training_state"], (self.opt_rest,))
        else:
            self.global_step = int(ck.get("global_step", 0))
        self._pending = None

    def _next_batch(self, stream_ctx):
        from .fused import prepare_train_rays
        cfg = self.config
        with stream_ctx:
            rays, ro, rd, rgb, fg, bg, t_min, t_max = prepare_train_rays(self.dataset, self.train_num_rays, self.gen,
                                                                         self.model, cfg["background_color"])
            handle = self.fused.march_begin(ro, rd, t_min, t_max)
            if self.fused.bg:  # the background's cone marching starts at the foreground box exit (t_max)
                handle = (handle, self.fused.bg_march_begin(ro, rd, t_max))
        return rays, rgb, fg, bg, handle

    def train_step(self):
        import contextlib
        from .parallel import all_reduce_gradients
        model, cfg, t = self.model, self.config, self.global_step
        model.update_step(0, t)  # cos anneal, progressive level / eps (and, on the reference's model, the refresh itself)
        grid = model.occupancy_grid
        refreshed = False
        if getattr(model, "refresh_owned_by_trainer", False) and cfg["grid_prune"] and t % 16 == 0:
            if self.device_occupancy_refresh:  # csrc/occupancy.hip: no host synchronisation, ~14 launches
                self.fused.refresh_occupancy_async(t, occ_thre=cfg.get("grid_prune_occ_thre", 0.01))
            else:
                grid.every_n_step(step=t, occ_eval_fn=self.fused.occ_eval_fn, occ_thre=cfg.get("grid_prune_occ_thre", 0.01))
            if self.fused.bg and self.device_occupancy_refresh:
                self.fused.refresh_bg_occupancy_async(t, occ_thre=cfg.get("grid_prune_occ_thre_bg", 0.01))
            elif self.fused.bg:
                model.occupancy_grid_bg.every_n_step(step=t, occ_eval_fn=self.fused.bg_occ_eval_fn,
                                                     occ_thre=cfg.get("grid_prune_occ_thre_bg", 0.01))
            refreshed = True
            if self.world_size > 1:  # DDP's broadcast_buffers: every rank goes on with rank 0's grids
                from .parallel import sync_occupancy_grid
                sync_occupancy_grid(grid)
                if self.fused.bg:
                    sync_occupancy_grid(model.occupancy_grid_bg)
        elif self.world_size > 1 and cfg["grid_prune"] and t % 16 == 0:
            #  the model refreshed its grids itself inside update_step (the reference's own model object): the ranks'
            # grids are synchronised here all the same
            from .parallel import sync_occupancy_grid
            sync_occupancy_grid(grid)
            if self.fused.bg:
                sync_occupancy_grid(model.occupancy_grid_bg)
        if refreshed or (cfg["grid_prune"] and t % 16 == 0):
            self._pending = None  # marched through the old grid
        main = torch.cuda.current_stream()
        if self._pending is None:
            self._pending = self._next_batch(contextlib.nullcontext())
            if self.fused.defer_bg_count:  # (the grids and their brick images are final on the main stream from here on)
                self._grids_ready = torch.cuda.Event()
                self._grids_ready.record(main)
        rays, rgb, fg, bg, handle = self._pending
        self._pending = None
        model.background_color = bg

        def after_march(n):
            if cfg["dynamic_ray_sampling"] and n > 0:  # systems/neus.py:93-95
                tr = int(self.train_num_rays * (self.train_num_samples / n))
                self.train_num_rays = min(int(self.train_num_rays * 0.9 + tr * 0.1), cfg["max_train_num_rays"])
            # the NEXT batch (its ray count is final now): ray preparation + marching on the side stream, underneath this
            # step's encode / networks / backward -- unless the next step refreshes the grid first
            if cfg["grid_prune"] and (t + 1) % 16 != 0:
                if self._side is None:
                    self._side = _shared_stream(self.device, "side")
                if self.fused.defer_bg_count:
                    # called when the whole step is queued (the background's kept count is read there): the batch must not
                    # wait for this step's kernels, only for the occupancy grids it marches through (refreshed on the main stream)
                    if self._grids_ready is not None:
                        self._side.wait_event(self._grids_ready)
                else:
                    ev = torch.cuda.Event()
                    ev.record(main)
                    self._side.wait_event(ev)
                self._pending = self._next_batch(torch.cuda.stream(self._side))
                for x in self._pending[:4]:
                    x.record_stream(main)

        scale = neus_lr_scale(t, self.config_name, self.max_steps)
        self.fused.table_adam = None
        if self.fuse_table_adam and self.sharded is None and self.world_size == 1:
            # AdamW on the hash tables inside their backward (csrc/hashgrid_owner.inc: OwnerAdam) at this step's learning rate;
            # the kernels take step count / bias corrections from the optimizer's device-side state
            self.fused.table_adam = {k: self.opt.table_update_desc(m, milestones=(), gamma=1.0, lr=self.opt.lr * scale)
                                     for k, m in self._table_of.items()}
        self.fused.table_bf16 = None
        if self.sharded is not None and self.sharded.transport == torch.bfloat16 and not os.environ.get("NSR_EXCHANGE_UNFUSED"):
            # multi-GPU: the table backwards write the exchange's bf16 send buffers themselves
            self.fused.table_bf16 = {k: self.sharded.send_buffer(m) for k, m in self._table_of.items()
                                     if self.sharded.state[m]["head"] == 0 and len(self.sharded.ranges(m)) == 1}
        guard = self._overflow_guard_state()
        if guard is not None:
            # Lightning's precision-16 protocol (the reference's NeuS configs train under it too) on the device, registered
            # around THIS step's launches: see _overflow_guard_state
            hs = float(self._guard_host[2:3].view(torch.float32)[0])  # (the scale as of one or two steps ago)
            if hs >= 1.0:
                self.fused.grad_scale = hs
            lib.nsr_overflow_guard(ctypes.c_void_p(guard.data_ptr()), float(self.fused.grad_scale))
        try:
            return self._finish_step(rays, rgb, fg, bg, handle, after_march, scale, guard)
        finally:
            if guard is not None:
                lib.nsr_overflow_guard(None, 0.0)
                self._guard_host.copy_(guard, non_blocking=True)

    def _overflow_guard_state(self):
        """int32[8] on the device, torch.cuda.amp.GradScaler's state where the kernels read it: {found-inf flag of even / odd
        steps, loss scale (float bits), clean steps, skipped steps, growth interval, -, -}.  A non-finite loss gradient (first
        kernel of the backward) or a non-finite gradient at the SDF network's output (an overflow of the fp16 colour network
        lands there) raises the step's flag; the tables' AdamW inside their backward, the networks' and the small tensors'
        optimizer launches then leave parameters, moments and step counts untouched, and the last of them halves the scale
        (x 2 again after ``overflow_growth_interval`` = 2,000 clean steps).  The scale is what the fp16 colour network's
        backward multiplies dL/dy by before rounding (``FusedNeuSStep.grad_scale``, 65,536): the host follows the device's
        value one or two steps late (an asynchronous read-back), so a run of overflowing steps may halve it once more than
        GradScaler would.  One GPU, table updates fused into the table backward; None: off (``NSR_NO_OVERFLOW_GUARD``)."""
        from .trainer import _GUARD_DEFAULT
        if not getattr(self, "overflow_guard", _GUARD_DEFAULT) or self.world_size > 1 or self.sharded is not None or \
                not self.fuse_table_adam:
            return None
        g = getattr(self, "_guard", None)
        if g is None:
            import struct
            bits = struct.unpack("<i", struct.pack("<f", float(self.fused.grad_scale)))[0]
            g = self._guard = torch.tensor([0, 0, bits, 0, 0, int(getattr(self, "overflow_growth_interval", 2000)), 0, 0],
                                           dtype=torch.int32, device=self.device)
            self._guard_host = torch.zeros(8, dtype=torch.int32).pin_memory()
        return g

    def overflow_guard_stats(self):
        """(loss scale, clean steps since its last change, skipped steps, optimizer steps taken) -- synchronises"""
        g = self._overflow_guard_state()
        if g is None:
            return None
        torch.cuda.synchronize(self.device)
        v = g.cpu()
        return {"scale": float(v[2:3].view(torch.float32)[0]), "clean_steps": int(v[3]), "skipped_steps": int(v[4]),
                "optimizer_steps": self.opt_rest.taken_steps()}

    def _finish_step(self, rays, rgb, fg, bg, handle, after_march, scale, guard):
        from .parallel import all_reduce_gradients
        model = self.model
        res = self.fused.forward_backward(rays, rgb, fg, bg, march_handle=handle, after_march=after_march)
        n = res["num_samples"]
        if "_bg_deferred" in res:
            # forward and backward are queued (> 1 ms of GPU work): if none of the step's polls has seen the background's kept
            # count yet, the host waits for it now and queues the next batch (its marching, side stream, ~0.25 ms, runs while
            # the host queues the optimizer steps below)
            self.fused.bg_count_deferred(res)
        if self.sharded is not None:
            for p in self._rest:  # every rank contributes the same tensor list (a rank may have marched nothing)
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
            all_reduce_gradients(self._rest)
            done = self.fused.bf16_written | self.fused.grad_written
            self.sharded.step(lr_scale=scale, timings=self.comm_timings, overwritten=self._tables,
                              prefilled=[self._table_of[k] for k in self.fused.bf16_written],
                              # a rank that marched nothing launched no table backward: its .grad is LAST step's
                              absent=[m for k, m in self._table_of.items() if k not in done])
        else:
            if self.world_size > 1:
                all_reduce_gradients(list(model.parameters()))
            self.opt.step(lr_scale=scale, updated_in_backward=[self._table_of[k] for k in self.fused.adam_applied])
        self.opt_rest.step(lr_scale=scale, device_step=guard is not None)  # (last: it updates the guard's scale)
        self.global_step += 1
        self.last = {"loss_acc": res["loss_acc"], "n_rays": rays.shape[0], "n_samples": n,
                     "n_samples_bg": res.get("num_samples_bg", 0), "n_marched_bg": res.get("num_marched_bg", 0)}
        return self.last
