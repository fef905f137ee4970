"""Parameter / occupancy holders for the fused step runners (``nsr.fused``, ``nsr.fused_neus``).

The fused runners do not own a model class: they read the hot-path tensors of ANY object that exposes them under the
reference's attribute paths -- the reference's own ``models.nerf.NeRFModel`` / ``models.neus.NeuSModel`` built on the
drop-in ``tinycudann`` / ``nerfacc`` packages, or the holder built here when no Lightning system is around
(``bench.py``, tools).  The holder is a *spec table*: every row is ``(state-dict path, factory)``; the rows are
instantiated into anonymous ``nn.Module`` containers so that ``state_dict()`` has exactly the reference's key set
(``geometry.encoding_with_network.params``, ``texture.network.params``, ``occupancy_grid._binary`` ... --
checkpoints of ``utils/mixins.py``/Lightning load with ``load_reference_checkpoint``) and nothing else of the
reference's class structure (no ``forward_``: rendering is the runners' job).
"""
import math

import torch
import torch.nn as nn

import tinycudann as tcnn
from nerfacc import ContractionType, OccupancyGrid
from nsr_hip import ops as _ops


def _attach(root, path, module):
    """root.a.b.c = module, creating anonymous containers on the way"""
    parts = path.split(".")
    cur = root
    for p in parts[:-1]:
        if not hasattr(cur, p):
            cur.add_module(p, nn.Module())
        cur = getattr(cur, p)
    cur.add_module(parts[-1], module)


class _Scalar(nn.Module):
    def __init__(self, name, value):
        super().__init__()
        self.register_parameter(name, nn.Parameter(torch.tensor(float(value))))


def _weight_normed_linear(d_in, d_out, weight_norm=True):
    """models/network_utils.py:131-139: the layer is wrapped in weight_norm only when the config asks for it"""
    lin = nn.Linear(d_in, d_out, bias=True)
    return nn.utils.weight_norm(lin) if weight_norm else lin


def sphere_init_(linear, first, last, radius, d_in, d_out):
    """geometric initialisation of the SDF network (reference models/network_utils.py:115-127); ``linear`` is the
    weight-normed layer: the direction parameter is written, then the magnitude is re-derived from it"""
    with torch.no_grad():
        w = linear.weight_v if hasattr(linear, "weight_v") else linear.weight
        if last:
            linear.bias.fill_(-radius)
            w.normal_(mean=math.sqrt(math.pi) / math.sqrt(d_in), std=0.0001)
        else:
            linear.bias.zero_()
            w.normal_(0.0, math.sqrt(2) / math.sqrt(d_out))
            if first:
                w[:, 3:].zero_()
        if hasattr(linear, "weight_g"):
            linear.weight_g.copy_(w.norm(dim=1, keepdim=True))


def _vanilla_head(n_in, n_out, cfg):
    """fp32 Linear stack of a reference ``VanillaMLP`` without sphere init (models/network_utils.py:128-131: zero bias,
    kaiming-uniform weights), as a container whose state-dict keys are ``layers.{0,2,4}.{weight,bias}``"""
    net, layers = nn.Module(), nn.Module()
    dims = [n_in] + [int(cfg["n_neurons"])] * int(cfg["n_hidden_layers"]) + [n_out]
    for i in range(len(dims) - 1):
        lin = nn.Linear(dims[i], dims[i + 1], bias=True)
        nn.init.constant_(lin.bias, 0.0)
        nn.init.kaiming_uniform_(lin.weight, nonlinearity="relu")
        layers.add_module(str(2 * i), lin)
    net.add_module("layers", layers)
    return net


def _colour_head(t):
    n_in = t["input_feature_dim"] + 16
    if t["mlp_network_config"]["otype"] == "VanillaMLP":
        return _vanilla_head(n_in, 3, t["mlp_network_config"])
    return tcnn.Network(n_in, 3, t["mlp_network_config"])


class HotPathState(nn.Module):
    """tensors of one scene: parameters (tcnn modules, the fp32 SDF head, the variance scalar), occupancy grid(s),
    scene box and the marching constants derived from the config"""

    def __init__(self, config):
        super().__init__()
        self.config = cfg = config
        kind = cfg["name"]
        r = float(cfg["radius"])
        self.register_buffer("scene_aabb", torch.tensor([-r, -r, -r, r, r, r], dtype=torch.float32))
        rows = []
        g, t = cfg["geometry"], cfg["texture"]
        sh = lambda: tcnn.Encoding(3, t["dir_encoding_config"])  # noqa: E731
        if kind == "nerf":
            rows += [("geometry.encoding_with_network",
                      lambda: tcnn.NetworkWithInputEncoding(3, g["feature_dim"], g["xyz_encoding_config"],
                                                            g["mlp_network_config"])),
                     ("texture.encoding.encoding", sh),
                     ("texture.network", lambda: _colour_head(t))]
        elif kind == "neus":
            enc_cfg = dict(g["xyz_encoding_config"])
            progressive = enc_cfg["otype"] == "ProgressiveBandHashGrid"
            enc_path = "geometry.encoding.encoding" + (".encoding" if progressive else "")
            n_enc = enc_cfg["n_levels"] * enc_cfg["n_features_per_level"] + (3 if enc_cfg.get("include_xyz") else 0)
            mc = g["mlp_network_config"]
            if mc["otype"] != "VanillaMLP" or mc["n_hidden_layers"] != 1 or mc["n_neurons"] != 64:
                raise NotImplementedError("fused NeuS state: the SDF head is the 1-hidden-layer fp32 VanillaMLP of the "
                                          "reference's neus-*/neuralangelo-* configs")
            wn = bool(mc.get("weight_norm", False))
            rows += [(enc_path, lambda: tcnn.Encoding(3, dict(enc_cfg, otype="HashGrid"))),
                     ("geometry.network.layers.0", lambda: _weight_normed_linear(n_enc, 64, wn)),
                     ("geometry.network.layers.2", lambda: _weight_normed_linear(64, g["feature_dim"], wn)),
                     ("texture.encoding.encoding", sh),
                     ("texture.network", lambda: _colour_head(t)),
                     ("variance", lambda: _Scalar("variance", cfg["variance"]["init_val"]))]
            self.progressive = dict(enc_cfg) if progressive else None
            self.current_level = enc_cfg.get("start_level", enc_cfg["n_levels"])
            self.grad_type = g["grad_type"]
            self.finite_difference_eps = None
            self.cos_anneal_ratio = 1.0
            if cfg.get("learned_background", False):  # NeRF++ branch: models/neus.py:52-59,69-74
                gb, tb = cfg["geometry_bg"], cfg["texture_bg"]
                eb = dict(gb["xyz_encoding_config"])
                if eb["otype"] != "HashGrid" or gb["mlp_network_config"]["otype"] != "VanillaMLP":
                    raise NotImplementedError("fused NeuS state: background = HashGrid + VanillaMLP density head")
                n_bg = eb["n_levels"] * eb["n_features_per_level"]
                rows += [("geometry_bg.encoding_with_network.encoding.encoding", lambda: tcnn.Encoding(3, eb)),
                         ("geometry_bg.encoding_with_network.network",
                          lambda: _vanilla_head(n_bg, gb["feature_dim"], gb["mlp_network_config"])),
                         ("texture_bg.encoding.encoding", lambda: tcnn.Encoding(3, tb["dir_encoding_config"])),
                         ("texture_bg.network", lambda: _colour_head(tb))]
                self.near_plane_bg, self.far_plane_bg = 0.1, 1e3
                self.cone_angle_bg = 10 ** (math.log10(self.far_plane_bg) / cfg["num_samples_per_ray_bg"]) - 1.0
                self.render_step_size_bg = 0.01
        else:
            raise ValueError(kind)
        for path, factory in rows:
            _attach(self, path, factory())
        if kind == "neus" and cfg["geometry"]["mlp_network_config"].get("sphere_init", False):
            rad = cfg["geometry"]["mlp_network_config"].get("sphere_init_radius", 0.5)
            l0, l2 = getattr(self.geometry.network.layers, "0"), getattr(self.geometry.network.layers, "2")
            sphere_init_(l0, True, False, rad, l0.in_features, 64)
            sphere_init_(l2, False, True, rad, 64, l2.out_features)
        self.contraction_type = ContractionType.AABB
        self.render_step_size = 1.732 * 2 * r / cfg["num_samples_per_ray"]  # models/nerf.py:33, models/neus.py:76
        if cfg["grid_prune"]:
            self.occupancy_grid = OccupancyGrid(roi_aabb=self.scene_aabb, resolution=128,
                                                contraction_type=ContractionType.AABB)
            if kind == "neus" and cfg.get("learned_background", False):
                self.occupancy_grid_bg = OccupancyGrid(roi_aabb=self.scene_aabb, resolution=256,
                                                       contraction_type=ContractionType.UN_BOUNDED_SPHERE)
        self.randomized = bool(cfg["randomized"])
        self.background_color = None
        self.schedules_restored = False  # set by update_step / restore_schedules (nsr.export refuses to guess the schedule)
        # the reference's NeuSModel refreshes its occupancy grid(s) inside update_step (models/neus.py:79-111); this holder
        # has no field code of its own, so the trainer drives the refresh through the fused runner's occ_eval_fn
        self.refresh_owned_by_trainer = kind == "neus"

    # -- schedules that live on the reference's model objects (update_step hooks) --------------------------------
    def update_step(self, epoch, global_step):
        cfg = self.config
        if cfg["name"] == "nerf" and self.training and cfg["grid_prune"]:
            self.occupancy_grid.every_n_step(step=global_step, occ_eval_fn=self.density_of_cells)
        self.restore_schedules(global_step)

    def restore_schedules(self, global_step):
        """the step-dependent state that lives on the reference's model objects and is NOT in the state dict -- cos-anneal
        ratio, progressive level, finite-difference eps (models/neus.py:86-87, models/network_utils.py:61-65,
        models/geometry.py:224-236).  The reference restores it through its batch-start hooks from the checkpoint's
        global_step; call this (or ``load_reference_checkpoint(..., global_step=...)``) before exporting from a loaded model."""
        cfg = self.config
        self.schedules_restored = True
        if cfg["name"] == "neus":
            end = cfg.get("cos_anneal_end", 0)
            self.cos_anneal_ratio = 1.0 if end == 0 else min(1.0, global_step / end)  # models/neus.py:86-87
            pg = self.progressive
            if pg is not None:  # models/network_utils.py:61-65, models/geometry.py:224-236
                lvl = min(pg["start_level"] + max(global_step - pg["start_step"], 0) // pg["update_steps"],
                          pg["n_levels"])
                self.current_level = max(self.current_level, lvl)
                if self.config["geometry"].get("finite_difference_eps") == "progressive":
                    res = pg["base_resolution"] * pg["per_level_scale"] ** (lvl - 1)
                    self.finite_difference_eps = 2 * float(cfg["radius"]) / res
            fd = self.config["geometry"].get("finite_difference_eps", 1e-3)
            if isinstance(fd, float):
                self.finite_difference_eps = fd

    def density_of_cells(self, x_world):
        """occupancy statistic of the nerf scene at world points (models/nerf.py:47-52): density x step size"""
        ewn = self.geometry.encoding_with_network
        x01 = _ops.contract_to_unisphere(x_world.float().contiguous(), float(self.config["radius"]),
                                         ContractionType.AABB.value)
        out = ewn(x01)
        dens, _ = _ops.density_activation(out.half().contiguous() if out.dtype != torch.float16 else out.contiguous(),
                                          out.shape[1], float(self.config["geometry"].get("density_bias", 0.0)),
                                          want_feature=False)
        return dens[..., None] * self.render_step_size

    def train(self, mode=True):
        self.randomized = bool(mode and self.config["randomized"])
        return super().train(mode)

    def load_reference_checkpoint(self, state_dict, strict=True, global_step=None):
        """a Lightning ``.ckpt`` of the reference stores the model under ``model.``: strip it and load.  ``state_dict`` may be
        the whole checkpoint (its ``state_dict`` / ``global_step`` entries are then used); ``global_step`` restores the
        step-dependent schedules (``restore_schedules``) the way the reference's hooks do after a resume."""
        if "state_dict" in state_dict and not any(torch.is_tensor(v) for v in state_dict.values()):
            global_step = state_dict.get("global_step", global_step) if global_step is None else global_step
            state_dict = state_dict["state_dict"]
        sd = {(k[len("model."):] if k.startswith("model.") else k): v for k, v in state_dict.items()}
        res = self.load_state_dict(sd, strict=strict)
        if global_step is not None:
            self.restore_schedules(int(global_step))
        return res


def build(config):
    return HotPathState(config)
