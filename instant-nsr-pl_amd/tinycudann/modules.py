"""``tinycudann.{Encoding, Network, NetworkWithInputEncoding}`` on libnsr_hip.so.

Contract kept from tiny-cuda-nn's torch bindings (the reference relies on every item):
  * each module is an ``nn.Module`` with EXACTLY ONE fp32 ``nn.Parameter`` named ``params``
    (``sphere_init_tcnn_network`` does ``list(network.parameters())[0].data`` and asserts its length,
    reference ``models/network_utils.py:155-156``);
  * attributes ``n_input_dims``, ``n_output_dims`` (read at ``network_utils.py:48,73``, ``texture.py:18``),
    ``dtype`` (fp16), ``seed``, ``loss_scale`` (128);
  * ``forward(x)``: any float dtype in, cast to fp32, returns fp16 ``[B, n_output_dims]``;
  * params are created on the CURRENT device (the reference constructs inside
    ``with torch.cuda.device(get_rank())``, ``network_utils.py:46,89,180,208``);
  * ``NetworkWithInputEncoding`` lays its flat params out as ``[network | encoding]``;
  * FullyFusedMLP weights: row-major ``[out,in]`` matrices concatenated, in/out padded to 16, padded
    inputs are 1.0 (``network_utils.py:142-173``); no biases;
  * unknown config keys are ignored (the reference leaks ``include_xyz``, ``start_level``,
    ``sphere_init``... into the dicts it passes, ``network_utils.py:90,181``);
  * encodings are double-differentiable w.r.t. their input (``models/geometry.py:177-180``).

Differences (documented in DESIGN.md): gradients accumulate in fp32 (no fp16 atomics), parameter init uses
torch's generator (tcnn's pcg32 stream is not reproducible), batch sizes need no padding to 128.
"""
import math

import torch

from nsr_hip import NsrError, make_grid_desc, make_mlp_desc
from nsr_hip import ops as _ops


def batch_size_granularity():
    """tcnn pads batches to 128/256; the gfx950 kernels take any batch size."""
    return 1


def free_temporary_memory():
    """tcnn frees its internal arena; libnsr_hip.so owns no device memory (reference models/utils.py:119)."""
    return None


def _current_device():
    if not torch.cuda.is_available():
        raise NsrError("tinycudann (MI355X build) needs a ROCm GPU: there is no CPU path in the product library")
    return torch.device("cuda", torch.cuda.current_device())


def _generator(seed, device):
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    return g


def _init_grid(desc, seed, device):
    n = desc.n_entries * desc.n_features
    return (torch.rand(n, generator=_generator(seed, device), device=device) * 2.0 - 1.0) * 1e-4


def _mlp_shapes(md):
    return [(64, md.in_pad)] + [(64, 64)] * (md.n_hidden - 1) + [(md.out_pad, 64)]


def _init_mlp(md, seed, device):
    g = _generator(seed, device)
    parts = []
    for o, i in _mlp_shapes(md):
        s = math.sqrt(6.0 / (o + i))
        parts.append((torch.rand(o * i, generator=g, device=device) * 2.0 - 1.0) * s)
    return torch.cat(parts)


def _grid_desc_from_config(cfg):
    otype = cfg.get("otype", "HashGrid")
    if otype not in ("HashGrid", "Grid") or str(cfg.get("type", "Hash")) != "Hash":
        raise NotImplementedError(f"tinycudann(gfx950): encoding otype={otype!r} type={cfg.get('type')!r} "
                                  "is not implemented (HashGrid, SphericalHarmonics only)")
    if str(cfg.get("interpolation", "Linear")) != "Linear":
        raise NotImplementedError("tinycudann(gfx950): only Linear interpolation is implemented")
    return make_grid_desc(cfg.get("n_levels", 16), cfg.get("n_features_per_level", 2),
                          cfg.get("log2_hashmap_size", 19), cfg.get("base_resolution", 16),
                          cfg.get("per_level_scale", 2.0))


def _mlp_desc_from_config(n_in, n_out, cfg):
    otype = cfg.get("otype", "FullyFusedMLP")
    if otype not in ("FullyFusedMLP", "CutlassMLP"):
        raise NotImplementedError(f"tinycudann(gfx950): network otype={otype!r} is not implemented")
    if int(cfg.get("n_neurons", 64)) != 64:
        raise NotImplementedError("tinycudann(gfx950): the fused MLP is 64 neurons wide (every reference config)")
    if str(cfg.get("activation", "ReLU")).lower() != "relu":
        raise NotImplementedError("tinycudann(gfx950): hidden activation must be ReLU")
    out_act = str(cfg.get("output_activation", "None")).lower()
    if out_act not in ("none", "sigmoid"):
        raise NotImplementedError(f"tinycudann(gfx950): output_activation={out_act!r} not implemented (None, Sigmoid)")
    md = make_mlp_desc(n_in, n_out, int(cfg.get("n_hidden_layers", 1)), out_act)
    if not (1 <= md.n_hidden <= 4) or md.in_pad > 64 or md.out_pad != 16:
        raise NotImplementedError(f"tinycudann(gfx950): unsupported MLP shape in={n_in} out={n_out} "
                                  f"hidden_layers={md.n_hidden} (in<=64, out<=16, 1..4 hidden layers)")
    return md


class Module(torch.nn.Module):
    """One flat fp32 parameter + a lazily refreshed fp16 shadow that the kernels read."""

    def __init__(self, seed=1337):
        super().__init__()
        self.seed = seed
        self.dtype = torch.float16
        self.loss_scale = 128.0
        dev = _current_device()
        self.params = torch.nn.Parameter(self._initial_params(seed, dev).to(torch.float32), requires_grad=True)
        self._shadow, self._shadow_key, self._shadow_trusted, self._shadow_lazy = None, None, False, False

    # ---- fp16 shadow (tcnn casts params to fp16 on every call) --------------------------------------------------
    def half_params(self, params):
        """fp16 copy of the parameters that the kernels read (tcnn casts its fp32 params on every call as well).  It is
        re-cast on EVERY call (75 MB of traffic, ~15 us on MI355X for the 12.6 M-parameter table) because writes through
        ``params.data`` -- the reference's ``sphere_init_tcnn_network`` (models/network_utils.py:155,172), DDP's initial
        broadcast -- bump no version counter and would otherwise be invisible.  The exception: a fused optimizer that writes
        the fp16 image itself in the pass that updates the parameters hands it over (``adopt_shadow``); it is then used as
        is until the parameter tensor is replaced or modified in place (version counter), or ``invalidate()`` is called."""
        key = (params.data_ptr(), params._version, params.device)
        # inference in eval() mode (chunked rendering, export) reuses a cast made in that mode for as long as the version key
        # holds -- one 75 MB cast per chunk otherwise; a `.data` write in between needs invalidate().  NOT in train() mode, not
        # even under no_grad: the reference's sphere_init writes `.data` right after construction, possibly behind a forward
        # (tests/test_gpu_shims.py::test_sphere_init_through_params_data_is_seen_and_trains)
        lazy_ok = not self.training
        if self._shadow is None or key != self._shadow_key or not (self._shadow_trusted or (lazy_ok and self._shadow_lazy)):
            self._shadow = params.detach().to(torch.float16).contiguous()
            self._shadow_key, self._shadow_trusted, self._shadow_lazy = key, False, lazy_ok
        return self._shadow

    def adopt_shadow(self, shadow):
        """``shadow`` IS the current fp16 image of ``params`` (written by the fused AdamW kernel in the same pass)"""
        p = self.params
        self._shadow, self._shadow_key, self._shadow_trusted = shadow, (p.data_ptr(), p._version, p.device), True

    def invalidate(self):
        """call after writing to ``params.data`` directly (``.data`` writes do not bump the version counter)"""
        self._shadow_key, self._shadow_trusted, self._shadow_lazy = None, False, False

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self.invalidate()  # (load_state_dict copies into params under no_grad; an adopted fp16 image is stale now)

    def _prep(self, x):
        if not x.is_cuda:
            raise NsrError("tinycudann(gfx950): input must be a GPU tensor (tcnn would warn and copy; we refuse)")
        if x.dim() != 2 or x.shape[1] != self.n_input_dims:
            raise ValueError(f"expected input [B, {self.n_input_dims}], got {tuple(x.shape)}")
        return x.to(torch.float32).contiguous()

    def extra_repr(self):
        return f"n_input_dims={self.n_input_dims}, n_output_dims={self.n_output_dims}, seed={self.seed}, " \
               f"dtype={self.dtype}, n_params={self.params.numel()}"


class _WithEmptyParams(torch.autograd.Function):
    """identity on ``y`` that makes the module's zero-element ``params`` part of the graph (its gradient: an empty tensor)"""

    @staticmethod
    def forward(ctx, y, params):
        ctx.shape = params.shape
        return y.view_as(y)

    @staticmethod
    def backward(ctx, g):
        return g, g.new_zeros(ctx.shape, dtype=torch.float32)


class Encoding(Module):
    def __init__(self, n_input_dims, encoding_config, seed=1337, dtype=None):
        self.n_input_dims = int(n_input_dims)
        self.encoding_config = dict(encoding_config)
        otype = self.encoding_config.get("otype")
        if otype == "SphericalHarmonics":
            if self.n_input_dims != 3 or int(self.encoding_config.get("degree", 4)) != 4:
                raise NotImplementedError("tinycudann(gfx950): SphericalHarmonics is implemented for degree 4, 3-D input")
            self.kind, self.grid_desc, self.n_output_dims = "sh", None, 16
        else:
            if self.n_input_dims != 3:
                raise NotImplementedError("tinycudann(gfx950): HashGrid is implemented for 3-D input")
            self.kind = "grid"
            self.grid_desc = _grid_desc_from_config(self.encoding_config)
            self.n_output_dims = self.grid_desc.n_levels * self.grid_desc.n_features
        super().__init__(seed)
        if dtype is not None:
            self.dtype = dtype

    def _initial_params(self, seed, dev):
        return _init_grid(self.grid_desc, seed, dev) if self.kind == "grid" else torch.zeros(0, device=dev)

    # ---- owner protocol of nsr_hip.ops._GridEncode ----
    def table_half(self, params):
        return self.half_params(params)

    def level_mask_count(self):
        return self.grid_desc.n_levels

    def grid_slice(self, flat):
        return flat

    def forward(self, x):
        x = self._prep(x)
        if self.kind == "sh":
            y = _ops.sh4_forward(x.detach())  # no gradient to directions (unused by the reference)
            if torch.is_grad_enabled() and self.params.requires_grad:
                # tcnn hands autograd a (zero-element) parameter gradient for parameter-free encodings too; DDP with
                # find_unused_parameters=False (reference launch.py:93-107) waits for one from EVERY parameter
                y = _WithEmptyParams.apply(y, self.params)
        else:
            y = _ops.grid_encode(x, self.params, self)
        return y if self.dtype == torch.float16 else y.to(self.dtype)


class Network(Module):
    def __init__(self, n_input_dims, n_output_dims, network_config, seed=1337):
        self.n_input_dims, self.n_output_dims = int(n_input_dims), int(n_output_dims)
        self.network_config = dict(network_config)
        self.mlp_desc = _mlp_desc_from_config(self.n_input_dims, self.n_output_dims, self.network_config)
        super().__init__(seed)

    def _initial_params(self, seed, dev):
        return _init_mlp(self.mlp_desc, seed, dev)

    def weights_half(self, params):
        return self.half_params(params)

    def mlp_slice(self, flat):
        return flat

    def forward(self, x):
        return _ops.mlp(self._prep(x), self.params, self)


class NetworkWithInputEncoding(Module):
    def __init__(self, n_input_dims, n_output_dims, encoding_config, network_config, seed=1337):
        self.n_input_dims, self.n_output_dims = int(n_input_dims), int(n_output_dims)
        self.encoding_config, self.network_config = dict(encoding_config), dict(network_config)
        if self.n_input_dims != 3:
            raise NotImplementedError("tinycudann(gfx950): HashGrid is implemented for 3-D input")
        self.grid_desc = _grid_desc_from_config(self.encoding_config)
        enc_dims = self.grid_desc.n_levels * self.grid_desc.n_features
        self.mlp_desc = _mlp_desc_from_config(enc_dims, self.n_output_dims, self.network_config)
        self.n_network_params = sum(o * i for o, i in _mlp_shapes(self.mlp_desc))
        super().__init__(seed)

    def _initial_params(self, seed, dev):
        return torch.cat([_init_mlp(self.mlp_desc, seed, dev), _init_grid(self.grid_desc, seed + 1, dev)])

    def table_half(self, params):
        return self.half_params(params)[self.n_network_params:]

    def weights_half(self, params):
        return self.half_params(params)[:self.n_network_params]

    def level_mask_count(self):
        return self.grid_desc.n_levels

    def grid_slice(self, flat):
        return None if flat is None else flat[self.n_network_params:]

    def mlp_slice(self, flat):
        return None if flat is None else flat[:self.n_network_params]

    def forward(self, x):
        return _ops.grid_mlp(self._prep(x), self.params, self)
