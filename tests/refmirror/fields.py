"""Fields of the hot path: density / SDF geometry and radiance texture (mirror of reference
``models/geometry.py:116-238``, ``models/texture.py:11-38``, ``models/network_utils.py:40-215``)."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

import tinycudann as tcnn
from nerfacc import ContractionType
from nsr_hip import ops as _ops


# ---------------------------------------------------------------------------------------------------------------
# activations
# ---------------------------------------------------------------------------------------------------------------
class _TruncExp(torch.autograd.Function):
    """exp in fp32; the gradient clamps the exponent at 15 (reference models/utils.py:53-68)"""

    @staticmethod
    def forward(ctx, x):
        x = x.float()
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        return g * torch.exp(torch.clamp(ctx.saved_tensors[0], max=15))


trunc_exp = _TruncExp.apply


def get_activation(name):
    """the subset of reference models/utils.py:71-97 that the hot-path configs use"""
    if name is None or str(name).lower() == "none":
        return lambda x: x
    name = str(name).lower()
    if name == "trunc_exp":
        return trunc_exp
    if name == "sigmoid":
        return torch.sigmoid
    if name.startswith("scale"):
        s = float(name[5:])
        return lambda x: x.clamp(0.0, s) / s
    if name.startswith("+") or name.startswith("-"):
        return lambda x: x + float(name)
    return getattr(F, name)


def contract_to_unisphere(x, radius, contraction_type):
    """reference models/geometry.py:17-29.  Positions that carry no gradient take the fused HIP kernel; the
    differentiable path (NeuS analytic normals need d/dx twice) stays in torch."""
    if not x.requires_grad and x.is_cuda and x.dtype == torch.float32:
        return _ops.contract_to_unisphere(x.reshape(-1, 3).contiguous(), radius, contraction_type.value).view(x.shape)
    x = (x + radius) / (2 * radius)
    if contraction_type == ContractionType.AABB:
        return x
    if contraction_type == ContractionType.UN_BOUNDED_SPHERE:
        x = x * 2 - 1
        mag = x.norm(dim=-1, keepdim=True)
        x = torch.where(mag > 1, (2 - 1 / mag) * (x / mag), x)
        return x / 4 + 0.5
    raise NotImplementedError(contraction_type)


# ---------------------------------------------------------------------------------------------------------------
# encodings / MLPs
# ---------------------------------------------------------------------------------------------------------------
class ProgressiveBandHashGrid(nn.Module):
    """levels >= current_level are masked to zero (reference models/network_utils.py:40-65).  The mask is applied
    inside the kernel (``level_mask_count``): masked levels are never gathered, instead of gathered then zeroed."""

    def __init__(self, in_channels, config):
        super().__init__()
        self.n_input_dims = in_channels
        cfg = dict(config, otype="HashGrid")
        self.encoding = tcnn.Encoding(in_channels, cfg)
        self.n_output_dims = self.encoding.n_output_dims
        self.n_level, self.n_features_per_level = cfg["n_levels"], cfg["n_features_per_level"]
        self.start_level, self.start_step, self.update_steps = cfg["start_level"], cfg["start_step"], cfg["update_steps"]
        self.current_level = self.start_level
        self.encoding.level_mask_count = lambda: self.current_level

    def forward(self, x):
        return self.encoding(x)

    def update_step(self, epoch, global_step):
        level = min(self.start_level + max(global_step - self.start_step, 0) // self.update_steps, self.n_level)
        self.current_level = max(level, self.current_level)  # the reference's mask is monotone (:65)


class VanillaFrequency(nn.Module):
    """NeRF positional encoding with an optional coarse-to-fine cosine mask (reference models/network_utils.py:14-37;
    pure torch there too -- the C1 CPU-plumbing encoder, kept double-differentiable).

    out = cat over k of [sin(2^k x) m_k, cos(2^k x) m_k];  without ``n_masking_step`` every m_k is 1, with it
    m_k = (1 - cos(pi * clamp(step / n_masking_step * n_freq - k, 0, 1))) / 2, refreshed by ``update_step``."""

    def __init__(self, in_channels, config):
        super().__init__()
        self.n_freq = int(config["n_frequencies"])
        self.n_input_dims = self.in_channels = int(in_channels)
        self.n_output_dims = self.in_channels * 2 * self.n_freq
        self.n_masking_step = int(config.get("n_masking_step", 0))
        self.freq_bands = 2.0 ** torch.linspace(0, self.n_freq - 1, self.n_freq)
        self.update_step(None, None)

    def forward(self, x):
        cols = []
        for k in range(self.n_freq):
            arg = self.freq_bands[k] * x
            cols += [torch.sin(arg) * self.mask[k], torch.cos(arg) * self.mask[k]]
        return torch.cat(cols, dim=-1)

    def update_step(self, epoch, global_step):
        if self.n_masking_step <= 0 or global_step is None:
            self.mask = torch.ones(self.n_freq, dtype=torch.float32)
        else:
            ramp = (global_step / self.n_masking_step * self.n_freq - torch.arange(0, self.n_freq)).clamp(0, 1)
            self.mask = (1.0 - torch.cos(math.pi * ramp)) / 2.0


class CompositeEncoding(nn.Module):
    """optionally prepends ``x * 2 - 1`` (reference models/network_utils.py:68-79)"""

    def __init__(self, encoding, include_xyz=False, xyz_scale=2.0, xyz_offset=-1.0):
        super().__init__()
        self.encoding = encoding
        self.include_xyz, self.xyz_scale, self.xyz_offset = include_xyz, xyz_scale, xyz_offset
        self.n_output_dims = int(include_xyz) * encoding.n_input_dims + encoding.n_output_dims

    def forward(self, x):
        e = self.encoding(x)
        return e if not self.include_xyz else torch.cat([x * self.xyz_scale + self.xyz_offset, e], dim=-1)

    def update_step(self, epoch, global_step):
        if hasattr(self.encoding, "update_step"):
            self.encoding.update_step(epoch, global_step)


def get_encoding(n_input_dims, config):
    """reference models/network_utils.py:82-92"""
    if config["otype"] == "VanillaFrequency":
        enc = VanillaFrequency(n_input_dims, config)
    elif config["otype"] == "ProgressiveBandHashGrid":
        enc = ProgressiveBandHashGrid(n_input_dims, config)
    else:
        enc = tcnn.Encoding(n_input_dims, config)
    return CompositeEncoding(enc, include_xyz=config.get("include_xyz", False), xyz_scale=2.0, xyz_offset=-1.0)


class VanillaMLP(nn.Module):
    """fp32 Linear stack with biases, sphere initialisation (softplus beta=100) and weight norm -- the SDF network of
    the NeuS configs (reference models/network_utils.py:95-139).  Stays in torch: it must be double-differentiable."""

    def __init__(self, dim_in, dim_out, config):
        super().__init__()
        self.n_neurons, self.n_hidden_layers = config["n_neurons"], config["n_hidden_layers"]
        self.sphere_init, self.weight_norm = config.get("sphere_init", False), config.get("weight_norm", False)
        self.sphere_init_radius = config.get("sphere_init_radius", 0.5)
        dims = [dim_in] + [self.n_neurons] * self.n_hidden_layers + [dim_out]
        layers = []
        for i in range(len(dims) - 1):
            layers.append(self._linear(dims[i], dims[i + 1], i == 0, i == len(dims) - 2))
            if i < len(dims) - 2:
                layers.append(nn.Softplus(beta=100) if self.sphere_init else nn.ReLU(inplace=True))
        self.layers = nn.Sequential(*layers)
        self.output_activation = get_activation(config.get("output_activation"))

    def _linear(self, d_in, d_out, first, last):
        layer = nn.Linear(d_in, d_out, bias=True)
        if self.sphere_init:
            if last:
                nn.init.constant_(layer.bias, -self.sphere_init_radius)
                nn.init.normal_(layer.weight, mean=math.sqrt(math.pi) / math.sqrt(d_in), std=0.0001)
            elif first:
                nn.init.constant_(layer.bias, 0.0)
                nn.init.constant_(layer.weight[:, 3:], 0.0)
                nn.init.normal_(layer.weight[:, :3], 0.0, math.sqrt(2) / math.sqrt(d_out))
            else:
                nn.init.constant_(layer.bias, 0.0)
                nn.init.normal_(layer.weight, 0.0, math.sqrt(2) / math.sqrt(d_out))
        else:
            nn.init.constant_(layer.bias, 0.0)
            nn.init.kaiming_uniform_(layer.weight, nonlinearity="relu")
        return nn.utils.weight_norm(layer) if self.weight_norm else layer

    def forward(self, x):
        with torch.autocast("cuda", enabled=False):
            return self.output_activation(self.layers(x.float()))


def get_mlp(n_input_dims, n_output_dims, config):
    """reference models/network_utils.py:176-184"""
    if config["otype"] == "VanillaMLP":
        return VanillaMLP(n_input_dims, n_output_dims, config)
    net = tcnn.Network(n_input_dims, n_output_dims, config)
    if config.get("sphere_init", False):
        raise NotImplementedError("sphere_init of a fused MLP is reference-side code (network_utils.py:142-173)")
    return net


class EncodingWithNetwork(nn.Module):
    def __init__(self, encoding, network):
        super().__init__()
        self.encoding, self.network = encoding, network

    def forward(self, x):
        return self.network(self.encoding(x))

    def update_step(self, epoch, global_step):
        self.encoding.update_step(epoch, global_step)


def get_encoding_with_network(n_input_dims, n_output_dims, encoding_config, network_config):
    """fused tcnn.NetworkWithInputEncoding unless a pure-torch piece is involved (network_utils.py:200-215)"""
    if encoding_config["otype"] in ("VanillaFrequency", "ProgressiveBandHashGrid") or \
            network_config["otype"] == "VanillaMLP":
        enc = get_encoding(n_input_dims, encoding_config)
        return EncodingWithNetwork(enc, get_mlp(enc.n_output_dims, n_output_dims, network_config))
    return tcnn.NetworkWithInputEncoding(n_input_dims=n_input_dims, n_output_dims=n_output_dims,
                                         encoding_config=encoding_config, network_config=network_config)


# ---------------------------------------------------------------------------------------------------------------
# fields
# ---------------------------------------------------------------------------------------------------------------
class VolumeDensity(nn.Module):
    """density + feature field (reference models/geometry.py:116-141)"""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.radius = config["radius"]
        self.contraction_type = None  # assigned by the renderer
        self.n_output_dims = config["feature_dim"]
        self.encoding_with_network = get_encoding_with_network(3, self.n_output_dims, config["xyz_encoding_config"],
                                                               config["mlp_network_config"])

    def forward(self, points):
        x = contract_to_unisphere(points, self.radius, self.contraction_type)
        out = self.encoding_with_network(x.view(-1, 3)).view(*x.shape[:-1], self.n_output_dims).float()
        density, feature = out[..., 0], out
        if "density_activation" in self.config:
            density = get_activation(self.config["density_activation"])(density + float(self.config["density_bias"]))
        if "feature_activation" in self.config:
            feature = get_activation(self.config["feature_activation"])(feature)
        return density, feature

    def update_step(self, epoch, global_step):
        if hasattr(self.encoding_with_network, "update_step"):
            self.encoding_with_network.update_step(epoch, global_step)


class VolumeSDF(nn.Module):
    """SDF + analytic / finite-difference gradient + feature (reference models/geometry.py:144-238)"""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.radius = config["radius"]
        self.contraction_type = None
        self.n_output_dims = config["feature_dim"]
        self.encoding = get_encoding(3, config["xyz_encoding_config"])
        self.network = get_mlp(self.encoding.n_output_dims, self.n_output_dims, config["mlp_network_config"])
        self.grad_type = config["grad_type"]
        self.finite_difference_eps = config.get("finite_difference_eps", 1e-3)
        self._finite_difference_eps = None

    def _sdf_feature(self, x01):
        out = self.network(self.encoding(x01.view(-1, 3))).view(*x01.shape[:-1], self.n_output_dims).float()
        sdf, feature = out[..., 0], out
        if "sdf_activation" in self.config:
            sdf = get_activation(self.config["sdf_activation"])(sdf + float(self.config["sdf_bias"]))
        if "feature_activation" in self.config:
            feature = get_activation(self.config["feature_activation"])(feature)
        return sdf, feature

    def forward(self, points, with_grad=True, with_feature=True, with_laplace=False):
        analytic = with_grad and self.grad_type == "analytic"
        with torch.set_grad_enabled(self.training or analytic):
            if analytic:
                if not self.training:
                    points = points.clone()
                points.requires_grad_(True)
            points_ = points
            x = contract_to_unisphere(points, self.radius, self.contraction_type)
            sdf, feature = self._sdf_feature(x)
            grad = laplace = None
            if with_grad and self.grad_type == "analytic":
                (grad,) = torch.autograd.grad(sdf, points_, grad_outputs=torch.ones_like(sdf), create_graph=True,
                                              retain_graph=True, only_inputs=True)
            elif with_grad:  # finite differences: 6 taps +-eps per axis (models/geometry.py:181-199)
                eps = self._finite_difference_eps
                offsets = torch.as_tensor([[eps, 0, 0], [-eps, 0, 0], [0, eps, 0], [0, -eps, 0], [0, 0, eps],
                                           [0, 0, -eps]], dtype=points_.dtype, device=points_.device)
                pd = (points_[..., None, :] + offsets).clamp(-self.radius, self.radius)
                pd = (pd + self.radius) / (2 * self.radius)  # plain AABB scaling, as the reference does (:194)
                sd = self.network(self.encoding(pd.view(-1, 3)))[..., 0].view(*points.shape[:-1], 6).float()
                grad = 0.5 * (sd[..., 0::2] - sd[..., 1::2]) / eps
                if with_laplace:
                    laplace = (sd[..., 0::2] + sd[..., 1::2] - 2 * sdf[..., None]).sum(-1) / (eps ** 2)
        rv = [sdf]
        if with_grad:
            rv.append(grad)
        if with_feature:
            rv.append(feature)
        if with_laplace:
            assert self.grad_type == "finite_difference", "Laplace needs grad_type='finite_difference'"
            rv.append(laplace)
        rv = [v if self.training else v.detach() for v in rv]
        return rv[0] if len(rv) == 1 else rv

    def update_step(self, epoch, global_step):
        self.encoding.update_step(epoch, global_step)
        if self.grad_type == "finite_difference":
            if isinstance(self.finite_difference_eps, float):
                self._finite_difference_eps = self.finite_difference_eps
            elif self.finite_difference_eps == "progressive":
                hg = self.config["xyz_encoding_config"]
                assert hg["otype"] == "ProgressiveBandHashGrid"
                level = min(hg["start_level"] + max(global_step - hg["start_step"], 0) // hg["update_steps"],
                            hg["n_levels"])
                grid_res = hg["base_resolution"] * hg["per_level_scale"] ** (level - 1)
                self._finite_difference_eps = 2 * self.radius / grid_res
            else:
                raise ValueError(f"Unknown finite_difference_eps={self.finite_difference_eps}")


class VolumeRadiance(nn.Module):
    """[feature | SH(dir) | extra(normal)] -> MLP -> rgb (reference models/texture.py:11-38)"""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.n_dir_dims, self.n_output_dims = config.get("n_dir_dims", 3), 3
        self.encoding = get_encoding(self.n_dir_dims, config["dir_encoding_config"])
        self.n_input_dims = config["input_feature_dim"] + self.encoding.n_output_dims
        self.network = get_mlp(self.n_input_dims, self.n_output_dims, config["mlp_network_config"])

    def forward(self, features, dirs, *args):
        dirs = (dirs + 1.0) / 2.0
        emb = self.encoding(dirs.view(-1, self.n_dir_dims))
        inp = torch.cat([features.view(-1, features.shape[-1]), emb] + [a.view(-1, a.shape[-1]) for a in args], dim=-1)
        color = self.network(inp).view(*features.shape[:-1], self.n_output_dims).float()
        if "color_activation" in self.config:
            color = get_activation(self.config["color_activation"])(color)
        return color

    def update_step(self, epoch, global_step):
        self.encoding.update_step(epoch, global_step)


class VarianceNetwork(nn.Module):
    """inv_s = exp(10 * variance) with an optional clamp schedule (reference models/neus.py:15-44)"""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.variance = nn.Parameter(torch.tensor(float(config["init_val"])))
        self.modulate = config.get("modulate", False)
        self.do_mod = False
        if self.modulate:
            self.mod_start_steps, self.reach_max_steps = config["mod_start_steps"], config["reach_max_steps"]
            self.max_inv_s = config["max_inv_s"]

    @property
    def inv_s(self):
        val = torch.exp(self.variance * 10.0)
        if self.modulate and self.do_mod:
            val = val.clamp_max(self.mod_val)
        return val

    def forward(self, x):
        return torch.ones([len(x), 1], device=self.variance.device) * self.inv_s

    def update_step(self, epoch, global_step):
        if self.modulate:
            self.do_mod = global_step > self.mod_start_steps
            if not self.do_mod:
                self.prev_inv_s = self.inv_s.item()
            else:
                self.mod_val = min((global_step / self.reach_max_steps) * (self.max_inv_s - self.prev_inv_s)
                                   + self.prev_inv_s, self.max_inv_s)
