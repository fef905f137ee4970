"""Oracle restatement of the tiny-cuda-nn pieces the reference calls.  TEST INFRASTRUCTURE ONLY.

tiny-cuda-nn is an UN-PINNED third-party dependency of the reference (``README.md:34``) whose source
is not under ``/root/reference``; this file restates its *published* algorithms in plain PyTorch (CPU,
fp32, differentiable to any order through autograd) and mirrors the Python surface the reference
touches, so ``models/network_utils.py`` can be imported on top of it unchanged:

* ``Encoding(n_input_dims, encoding_config)``      <- ``models/network_utils.py:47,90``
* ``Network(n_input_dims, n_output_dims, cfg)``    <- ``models/network_utils.py:181``
* ``NetworkWithInputEncoding(...)`` (kwargs)       <- ``models/network_utils.py:209-214``
* ``free_temporary_memory()``                      <- ``models/utils.py:119``

Algorithms:
  A.1 multiresolution hash grid, linear interpolation, coherent prime hash (Instant-NGP sec. 3).
  A.2 real spherical harmonics, degree 4 (16 coefficients), input mapped [0,1]^3 -> [-1,1]^3.
  A.3 FullyFusedMLP: 64-wide, no bias, fp16 weights/activations, in/out padded to 16, padded inputs
      are the constant 1.0 (layout corroborated by ``models/network_utils.py:142-173``).

Numerics of this oracle (normative for the HIP path's parity tests):
  * hash grid: table values rounded to fp16, trilinear blend accumulated in fp32, output rounded to
    fp16 ONCE (tcnn accumulates in fp16; we are strictly more accurate, tolerance in tests).
  * MLP: weights and inter-layer activations rounded to fp16, fp32 accumulation, output activation
    evaluated in fp32 and rounded to fp16 once.
  * level geometry (scale / resolution / offsets) is computed in **fp32 with glibc log2f/exp2f**, the
    same calls the product's host-side C code makes (``nsr_hashgrid_make_desc``).

PARITY STATUS: unpinned (no reference-owned vectors exist); pinned by KATs in tests/test_oracle_kat.py.
"""
import ctypes
import ctypes.util
import math

import numpy as np
import torch

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.log2f.restype = ctypes.c_float
_libm.log2f.argtypes = [ctypes.c_float]
_libm.exp2f.restype = ctypes.c_float
_libm.exp2f.argtypes = [ctypes.c_float]
_libm.ceilf.restype = ctypes.c_float
_libm.ceilf.argtypes = [ctypes.c_float]

PRIME_Y = 2654435761
PRIME_Z = 805459861
BATCH_GRANULARITY = 128


# ----------------------------------------------------------------------------------------------
# A.1  hash grid
# ----------------------------------------------------------------------------------------------
class GridDesc:
    """Per-level geometry of a tcnn ``HashGrid`` (all fp32 on the host)."""

    def __init__(self, n_levels, n_features_per_level, log2_hashmap_size, base_resolution, per_level_scale):
        self.L = int(n_levels)
        self.F = int(n_features_per_level)
        self.log2_T = int(log2_hashmap_size)
        self.base_resolution = int(base_resolution)
        self.per_level_scale = float(per_level_scale)
        log2s = _libm.log2f(np.float32(per_level_scale))
        f32 = np.float32
        self.scale, self.res, self.size, self.offset = [], [], [], [0]
        for l in range(self.L):
            # scale_l = exp2f(l * log2s) * base - 1  (every op rounded to fp32)
            e = _libm.exp2f(f32(f32(l) * f32(log2s)))
            scale = f32(f32(e) * f32(self.base_resolution)) - f32(1.0)
            res = int(_libm.ceilf(f32(scale))) + 1
            n = min(res ** 3, 2 ** 32 - 1)
            n = (n + 7) // 8 * 8
            n = min(n, 1 << self.log2_T)
            self.scale.append(float(f32(scale)))
            self.res.append(res)
            self.size.append(n)
            self.offset.append(self.offset[-1] + n)
        self.n_entries = self.offset[-1]
        self.n_params = self.n_entries * self.F
        self.n_output_dims = self.L * self.F

    @classmethod
    def from_config(cls, cfg):
        return cls(cfg["n_levels"], cfg["n_features_per_level"], cfg["log2_hashmap_size"],
                   cfg["base_resolution"], cfg.get("per_level_scale", 2.0))


def coherent_prime_hash(cx, cy, cz):
    """uint32 wrap-around xor hash of integer corner coordinates (int64 tensors or python ints)."""
    m = 0xFFFFFFFF
    return ((cx * 1) & m) ^ ((cy * PRIME_Y) & m) ^ ((cz * PRIME_Z) & m)


def grid_index(cx, cy, cz, res, size):
    """tcnn ``grid_index``: dense x-fastest while the stride fits, hashed otherwise, then ``% size``."""
    stride, index = 1, 0
    for c in (cx, cy, cz):
        if stride <= size:
            index = index + c * stride
            stride *= res
    if size < stride:
        index = coherent_prime_hash(cx, cy, cz)
    return index % size


def hashgrid_encode(x, table, desc, fp16=True):
    """x [N,3] fp32 in [0,1]; table [n_entries, F] fp32 (the module's flat params reshaped).

    Returns [N, L*F] fp32 holding fp16-representable values when ``fp16`` (level-major columns).
    Differentiable w.r.t. ``x`` (through the interpolation weights) and ``table`` to any order.
    """
    assert x.shape[-1] == 3
    if fp16:
        table = table.half().float()
    outs = []
    for l in range(desc.L):
        scale, res, size, off = desc.scale[l], desc.res[l], desc.size[l], desc.offset[l]
        # pos = fmaf(scale, x, 0.5f): product of two fp32 is exact in fp64
        pos = (x.double() * scale + 0.5).float()
        g = torch.floor(pos.detach())
        w = pos - g
        gi = g.to(torch.int64)
        # all 8 corners in ONE gather per level (one dense scatter in backward instead of eight)
        idxs, wgts = [], []
        for c in range(8):
            wgt = None
            corner = []
            for d in range(3):
                bit = (c >> d) & 1
                wd = w[:, d] if bit else (1.0 - w[:, d])
                wgt = wd if wgt is None else wgt * wd
                corner.append(gi[:, d] + bit)
            idxs.append(grid_index(corner[0], corner[1], corner[2], res, size))
            wgts.append(wgt)
        idx = torch.stack(idxs, dim=1)                      # [N, 8]
        wgt = torch.stack(wgts, dim=1)                      # [N, 8]
        feat = table[off + idx]                             # [N, 8, F]
        acc = (wgt[..., None] * feat).sum(dim=1)
        outs.append(acc)
    y = torch.cat(outs, dim=-1)
    if fp16:
        y = y.half().float()
    return y


# ----------------------------------------------------------------------------------------------
# A.2  spherical harmonics, degree 4
# ----------------------------------------------------------------------------------------------
def sh4_encode(u, fp16=True):
    """u [N,3] in [0,1] (the reference maps dirs with (d+1)/2 first, ``models/texture.py:24``)."""
    v = u * 2.0 - 1.0
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    xy, xz, yz = x * y, x * z, y * z
    x2, y2, z2 = x * x, y * y, z * z
    out = [
        torch.full_like(x, 0.28209479177387814),
        -0.48860251190291987 * y,
        0.48860251190291987 * z,
        -0.48860251190291987 * x,
        1.0925484305920792 * xy,
        -1.0925484305920792 * yz,
        0.94617469575755997 * z2 - 0.31539156525251999,
        -1.0925484305920792 * xz,
        0.54627421529603959 * (x2 - y2),
        0.59004358992664352 * y * (-3.0 * x2 + y2),
        2.8906114426405538 * xy * z,
        0.45704579946446572 * y * (1.0 - 5.0 * z2),
        0.3731763325901154 * z * (5.0 * z2 - 3.0),
        0.45704579946446572 * x * (1.0 - 5.0 * z2),
        1.4453057213202769 * z * (x2 - y2),
        0.59004358992664352 * x * (-x2 + 3.0 * y2),
    ]
    o = torch.stack(out, dim=-1)
    if fp16:
        o = o.half().float()
    return o


# ----------------------------------------------------------------------------------------------
# A.3  FullyFusedMLP
# ----------------------------------------------------------------------------------------------
def _pad16(n):
    return (n + 15) // 16 * 16


class MLPDesc:
    def __init__(self, n_input_dims, n_output_dims, cfg):
        self.n_in, self.n_out = int(n_input_dims), int(n_output_dims)
        self.in_pad, self.out_pad = _pad16(self.n_in), _pad16(self.n_out)
        self.width = int(cfg.get("n_neurons", 64))
        self.n_hidden = int(cfg.get("n_hidden_layers", 1))
        self.activation = str(cfg.get("activation", "ReLU")).lower()
        self.output_activation = str(cfg.get("output_activation", "None")).lower()
        assert self.width == 64, "oracle restates the 64-wide FullyFusedMLP only"
        assert self.activation == "relu"
        assert self.output_activation in ("none", "sigmoid")
        assert self.n_hidden >= 1
        self.shapes = [(self.width, self.in_pad)] + [(self.width, self.width)] * (self.n_hidden - 1) \
            + [(self.out_pad, self.width)]
        self.n_params = sum(o * i for o, i in self.shapes)

    def split(self, params):
        ws, off = [], 0
        for o, i in self.shapes:
            ws.append(params[off:off + o * i].view(o, i))
            off += o * i
        return ws


def mlp_forward(x, params, desc, fp16=True, return_padded=False):
    """x [N, n_in] fp32; params flat fp32 (row-major [out,in] matrices concatenated)."""
    q = (lambda t: t.half().float()) if fp16 else (lambda t: t)
    n = x.shape[0]
    if desc.in_pad > desc.n_in:
        x = torch.cat([x, torch.ones(n, desc.in_pad - desc.n_in, dtype=x.dtype)], dim=-1)
    h = q(x)
    ws = desc.split(params)
    for w in ws[:-1]:
        h = q(torch.relu(h @ q(w).t()))
    o = h @ q(ws[-1]).t()
    if desc.output_activation == "sigmoid":
        o = torch.sigmoid(o)
    o = q(o)
    return o if return_padded else o[:, :desc.n_out]


# ----------------------------------------------------------------------------------------------
# module surface (what the reference constructs)
# ----------------------------------------------------------------------------------------------
def _make_generator(seed):
    g = torch.Generator()
    g.manual_seed(int(seed))
    return g


def init_grid_params(desc, seed):
    """tcnn initialises the table U(-1e-4, 1e-4) (pcg32; not bit-reproducible -- tests load weights)."""
    g = _make_generator(seed)
    return (torch.rand(desc.n_params, generator=g) * 2.0 - 1.0) * 1e-4


def init_mlp_params(desc, seed):
    """Xavier-uniform per (padded) matrix."""
    g = _make_generator(seed)
    parts = []
    for o, i in desc.shapes:
        s = math.sqrt(6.0 / (o + i))
        parts.append((torch.rand(o * i, generator=g) * 2.0 - 1.0) * s)
    return torch.cat(parts)


class _Module(torch.nn.Module):
    dtype = torch.float16
    loss_scale = 128.0

    def __init__(self, seed=1337):
        super().__init__()
        self.seed = seed
        self.params = torch.nn.Parameter(self._initial_params(seed).float(), requires_grad=True)

    def forward(self, x):
        x = x.to(torch.float32).contiguous()
        y = self._forward(x, self.params)
        return y.to(self.dtype)[:, :self.n_output_dims]


class Encoding(_Module):
    def __init__(self, n_input_dims, encoding_config, seed=1337, dtype=None):
        self.n_input_dims = n_input_dims
        self.encoding_config = dict(encoding_config)
        otype = self.encoding_config["otype"]
        if otype in ("HashGrid", "Grid"):
            assert n_input_dims == 3
            self.desc = GridDesc.from_config(self.encoding_config)
            self.n_output_dims = self.desc.n_output_dims
            self.kind = "grid"
        elif otype == "SphericalHarmonics":
            assert n_input_dims == 3 and int(self.encoding_config.get("degree", 4)) == 4
            self.n_output_dims = 16
            self.kind = "sh"
        else:
            raise RuntimeError(f"oracle: unsupported encoding otype {otype}")
        super().__init__(seed)
        if dtype is not None:
            self.dtype = dtype

    def _initial_params(self, seed):
        return init_grid_params(self.desc, seed) if self.kind == "grid" else torch.zeros(0)

    def _forward(self, x, params):
        if self.kind == "grid":
            return hashgrid_encode(x, params.view(-1, self.desc.F), self.desc)
        return sh4_encode(x)


class Network(_Module):
    def __init__(self, n_input_dims, n_output_dims, network_config, seed=1337):
        self.n_input_dims, self.n_output_dims = n_input_dims, n_output_dims
        self.network_config = dict(network_config)
        self.desc = MLPDesc(n_input_dims, n_output_dims, self.network_config)
        super().__init__(seed)

    def _initial_params(self, seed):
        return init_mlp_params(self.desc, seed)

    def _forward(self, x, params):
        return mlp_forward(x, params, self.desc)


class NetworkWithInputEncoding(_Module):
    """Flat params = [network | encoding] (tcnn sets the network's slice first)."""

    def __init__(self, n_input_dims, n_output_dims, encoding_config, network_config, seed=1337):
        self.n_input_dims, self.n_output_dims = n_input_dims, n_output_dims
        self.encoding_config, self.network_config = dict(encoding_config), dict(network_config)
        self.grid = GridDesc.from_config(self.encoding_config)
        self.desc = MLPDesc(self.grid.n_output_dims, n_output_dims, self.network_config)
        super().__init__(seed)

    def _initial_params(self, seed):
        return torch.cat([init_mlp_params(self.desc, seed), init_grid_params(self.grid, seed + 1)])

    def _forward(self, x, params):
        n_net = self.desc.n_params
        enc = hashgrid_encode(x, params[n_net:].view(-1, self.grid.F), self.grid)
        return mlp_forward(enc, params[:n_net], self.desc)


def free_temporary_memory():
    return None
