"""Renderers of the hot path: march -> evaluate fields -> composite (mirror of reference ``models/nerf.py:15-127``
and ``models/neus.py:47-297``).  Output dictionaries carry the same keys / shapes / dtypes as the reference's."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from nerfacc import (ContractionType, OccupancyGrid, accumulate_along_rays, ray_marching, render_weight_from_alpha,
                     render_weight_from_density)
from nerfacc.intersection import ray_aabb_intersect
from nsr_hip import ops as _ops

from .fields import VarianceNetwork, VolumeDensity, VolumeRadiance, VolumeSDF


def _positions(rays_o, rays_d, ray_indices, t_starts, t_ends):
    """o[r] + d[r] * (t0+t1)/2 and the per-sample directions, one fused gather kernel
    (reference models/nerf.py:66-69,95-99)"""
    return _ops.sample_positions(rays_o, rays_d, ray_indices, t_starts.contiguous(), t_ends.contiguous())


def chunk_batch(func, chunk_size, move_to_cpu, *args, **kwargs):
    """evaluation-time chunking (reference models/utils.py:13-50): tensor / tuple / list / dict results, chunks
    detached when grad mode is off and optionally moved to the CPU; ``None`` chunks are skipped"""
    B = next(a.shape[0] for a in args if isinstance(a, torch.Tensor))
    out, out_type, length = {}, None, 0
    for i in range(0, B, chunk_size):
        o = func(*[a[i:i + chunk_size] if isinstance(a, torch.Tensor) else a for a in args], **kwargs)
        if o is None:
            continue
        out_type = type(o)
        if isinstance(o, torch.Tensor):
            o = {0: o}
        elif isinstance(o, (tuple, list)):
            length = len(o)
            o = dict(enumerate(o))
        elif not isinstance(o, dict):
            raise TypeError(f"chunk_batch: unsupported return type {type(o)}")
        for k, v in o.items():
            v = v if torch.is_grad_enabled() else v.detach()
            out.setdefault(k, []).append(v.cpu() if move_to_cpu else v)
    if out_type is None:
        return None
    out = {k: torch.cat(v, dim=0) for k, v in out.items()}
    if out_type is torch.Tensor:
        return out[0]
    if out_type in (tuple, list):
        return out_type([out[i] for i in range(length)])
    return out


class NeRFModel(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.geometry = VolumeDensity(config["geometry"])
        self.texture = VolumeRadiance(config["texture"])
        r = config["radius"]
        self.register_buffer("scene_aabb", torch.as_tensor([-r, -r, -r, r, r, r], dtype=torch.float32))
        if config["learned_background"]:
            self.occupancy_grid_res = 256
            self.near_plane, self.far_plane = 0.2, 1e4
            self.cone_angle = 10 ** (math.log10(self.far_plane) / config["num_samples_per_ray"]) - 1.0
            self.render_step_size = 0.01
            self.contraction_type = ContractionType.UN_BOUNDED_SPHERE
        else:
            self.occupancy_grid_res = 128
            self.near_plane, self.far_plane = None, None
            self.cone_angle = 0.0
            self.render_step_size = 1.732 * 2 * r / config["num_samples_per_ray"]
            self.contraction_type = ContractionType.AABB
        self.geometry.contraction_type = self.contraction_type
        if config["grid_prune"]:
            self.occupancy_grid = OccupancyGrid(roi_aabb=self.scene_aabb, resolution=self.occupancy_grid_res,
                                                contraction_type=self.contraction_type)
        self.randomized = config["randomized"]
        self.background_color = None

    def update_step(self, epoch, global_step):
        self.geometry.update_step(epoch, global_step)
        self.texture.update_step(epoch, global_step)

        def occ_eval_fn(x):
            density, _ = self.geometry(x)
            return density[..., None] * self.render_step_size  # ~ 1 - exp(-density * step)

        if self.training and self.config["grid_prune"]:
            self.occupancy_grid.every_n_step(step=global_step, occ_eval_fn=occ_eval_fn)

    def forward_(self, rays):
        n_rays = rays.shape[0]
        rays_o, rays_d = rays[:, 0:3].contiguous(), rays[:, 3:6].contiguous()

        def sigma_fn(t_starts, t_ends, ray_indices):
            positions, _ = _ops.sample_positions(rays_o, rays_d, ray_indices, t_starts, t_ends, want_dirs=False)
            density, _ = self.geometry(positions)
            return density[..., None]

        with torch.no_grad():
            ray_indices, t_starts, t_ends = ray_marching(
                rays_o, rays_d, scene_aabb=None if self.config["learned_background"] else self.scene_aabb,
                grid=self.occupancy_grid if self.config["grid_prune"] else None, sigma_fn=sigma_fn,
                near_plane=self.near_plane, far_plane=self.far_plane, render_step_size=self.render_step_size,
                stratified=self.randomized, cone_angle=self.cone_angle, alpha_thre=0.0)
        positions, t_dirs = _positions(rays_o, rays_d, ray_indices, t_starts, t_ends)
        midpoints = (t_starts + t_ends) / 2.0
        intervals = t_ends - t_starts
        density, feature = self.geometry(positions)
        rgb = self.texture(feature, t_dirs)
        weights = render_weight_from_density(t_starts, t_ends, density[..., None], ray_indices=ray_indices, n_rays=n_rays)
        opacity = accumulate_along_rays(weights, ray_indices, values=None, n_rays=n_rays)
        depth = accumulate_along_rays(weights, ray_indices, values=midpoints, n_rays=n_rays)
        comp_rgb = accumulate_along_rays(weights, ray_indices, values=rgb, n_rays=n_rays)
        comp_rgb = comp_rgb + self.background_color * (1.0 - opacity)
        out = {"comp_rgb": comp_rgb, "opacity": opacity, "depth": depth, "rays_valid": opacity > 0,
               "num_samples": torch.as_tensor([len(t_starts)], dtype=torch.int32, device=rays.device)}
        if self.training:
            out.update({"weights": weights.view(-1), "points": midpoints.view(-1), "intervals": intervals.view(-1),
                        "ray_indices": ray_indices.view(-1)})
        return out

    def forward(self, rays):
        if self.training:
            return self.forward_(rays)
        return chunk_batch(self.forward_, self.config["ray_chunk"], True, rays)

    def train(self, mode=True):
        self.randomized = mode and self.config["randomized"]
        return super().train(mode=mode)

    def eval(self):
        self.randomized = False
        return super().eval()


class NeuSModel(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.geometry = VolumeSDF(config["geometry"])
        self.texture = VolumeRadiance(dict(config["texture"]))
        self.geometry.contraction_type = ContractionType.AABB
        if config["learned_background"]:
            self.geometry_bg = VolumeDensity(config["geometry_bg"])
            self.texture_bg = VolumeRadiance(config["texture_bg"])
            self.geometry_bg.contraction_type = ContractionType.UN_BOUNDED_SPHERE
            self.near_plane_bg, self.far_plane_bg = 0.1, 1e3
            self.cone_angle_bg = 10 ** (math.log10(self.far_plane_bg) / config["num_samples_per_ray_bg"]) - 1.0
            self.render_step_size_bg = 0.01
        self.variance = VarianceNetwork(config["variance"])
        r = config["radius"]
        self.register_buffer("scene_aabb", torch.as_tensor([-r, -r, -r, r, r, r], dtype=torch.float32))
        if config["grid_prune"]:
            self.occupancy_grid = OccupancyGrid(roi_aabb=self.scene_aabb, resolution=128,
                                                contraction_type=ContractionType.AABB)
            if config["learned_background"]:
                self.occupancy_grid_bg = OccupancyGrid(roi_aabb=self.scene_aabb, resolution=256,
                                                       contraction_type=ContractionType.UN_BOUNDED_SPHERE)
        self.randomized = config["randomized"]
        self.background_color = None
        self.render_step_size = 1.732 * 2 * r / config["num_samples_per_ray"]
        self.cos_anneal_ratio = 1.0

    def update_step(self, epoch, global_step):
        self.geometry.update_step(epoch, global_step)
        self.texture.update_step(epoch, global_step)
        if self.config["learned_background"]:
            self.geometry_bg.update_step(epoch, global_step)
            self.texture_bg.update_step(epoch, global_step)
        self.variance.update_step(epoch, global_step)
        end = self.config.get("cos_anneal_end", 0)
        self.cos_anneal_ratio = 1.0 if end == 0 else min(1.0, global_step / end)

        def occ_eval_fn(x):  # closed-form alpha of one step at a flat SDF (reference models/neus.py:90-101)
            sdf = self.geometry(x, with_grad=False, with_feature=False)
            inv_s = self.variance.inv_s.detach().reshape(1, 1).clip(1e-6, 1e6)
            prev_cdf = torch.sigmoid((sdf[..., None] + self.render_step_size * 0.5) * inv_s)
            next_cdf = torch.sigmoid((sdf[..., None] - self.render_step_size * 0.5) * inv_s)
            return (((prev_cdf - next_cdf) + 1e-5) / (prev_cdf + 1e-5)).view(-1, 1).clip(0.0, 1.0)

        def occ_eval_fn_bg(x):
            density, _ = self.geometry_bg(x)
            return density[..., None] * self.render_step_size_bg

        if self.training and self.config["grid_prune"]:
            self.occupancy_grid.every_n_step(step=global_step, occ_eval_fn=occ_eval_fn,
                                             occ_thre=self.config.get("grid_prune_occ_thre", 0.01))
            if self.config["learned_background"]:
                self.occupancy_grid_bg.every_n_step(step=global_step, occ_eval_fn=occ_eval_fn_bg,
                                                    occ_thre=self.config.get("grid_prune_occ_thre_bg", 0.01))

    def get_alpha(self, sdf, normal, dirs, dists):
        """fused SDF->alpha kernel with its own backward (reference models/neus.py:117-139)"""
        return _ops.neus_alpha(sdf, normal, dirs, dists, self.variance.inv_s, self.cos_anneal_ratio)

    def forward_bg_(self, rays):
        n_rays = rays.shape[0]
        rays_o, rays_d = rays[:, 0:3].contiguous(), rays[:, 3:6].contiguous()

        def sigma_fn(t_starts, t_ends, ray_indices):
            positions, _ = _ops.sample_positions(rays_o, rays_d, ray_indices, t_starts, t_ends, want_dirs=False)
            density, _ = self.geometry_bg(positions)
            return density[..., None]

        _, t_max = ray_aabb_intersect(rays_o, rays_d, self.scene_aabb)
        near_plane = torch.where(t_max > 1e9, self.near_plane_bg, t_max)  # start at the foreground box exit
        with torch.no_grad():
            ray_indices, t_starts, t_ends = ray_marching(
                rays_o, rays_d, scene_aabb=None, grid=self.occupancy_grid_bg if self.config["grid_prune"] else None,
                sigma_fn=sigma_fn, near_plane=near_plane, far_plane=self.far_plane_bg,
                render_step_size=self.render_step_size_bg, stratified=self.randomized, cone_angle=self.cone_angle_bg,
                alpha_thre=0.0)
        positions, t_dirs = _positions(rays_o, rays_d, ray_indices, t_starts, t_ends)
        midpoints = (t_starts + t_ends) / 2.0
        density, feature = self.geometry_bg(positions)
        rgb = self.texture_bg(feature, t_dirs)
        weights = render_weight_from_density(t_starts, t_ends, density[..., None], ray_indices=ray_indices, n_rays=n_rays)
        opacity = accumulate_along_rays(weights, ray_indices, values=None, n_rays=n_rays)
        depth = accumulate_along_rays(weights, ray_indices, values=midpoints, n_rays=n_rays)
        comp_rgb = accumulate_along_rays(weights, ray_indices, values=rgb, n_rays=n_rays)
        comp_rgb = comp_rgb + self.background_color * (1.0 - opacity)
        out = {"comp_rgb": comp_rgb, "opacity": opacity, "depth": depth, "rays_valid": opacity > 0,
               "num_samples": torch.as_tensor([len(t_starts)], dtype=torch.int32, device=rays.device)}
        if self.training:
            out.update({"weights": weights.view(-1), "points": midpoints.view(-1),
                        "intervals": (t_ends - t_starts).view(-1), "ray_indices": ray_indices.view(-1)})
        return out

    def forward_(self, rays):
        n_rays = rays.shape[0]
        rays_o, rays_d = rays[:, 0:3].contiguous(), rays[:, 3:6].contiguous()
        with torch.no_grad():
            ray_indices, t_starts, t_ends = ray_marching(
                rays_o, rays_d, scene_aabb=self.scene_aabb,
                grid=self.occupancy_grid if self.config["grid_prune"] else None, alpha_fn=None, near_plane=None,
                far_plane=None, render_step_size=self.render_step_size, stratified=self.randomized, cone_angle=0.0,
                alpha_thre=0.0)
        positions, t_dirs = _positions(rays_o, rays_d, ray_indices, t_starts, t_ends)
        midpoints = (t_starts + t_ends) / 2.0
        dists = t_ends - t_starts
        fd = self.config["geometry"]["grad_type"] == "finite_difference"
        if fd:
            sdf, sdf_grad, feature, sdf_laplace = self.geometry(positions, with_grad=True, with_feature=True,
                                                                with_laplace=True)
        else:
            sdf, sdf_grad, feature = self.geometry(positions, with_grad=True, with_feature=True)
        normal = F.normalize(sdf_grad, p=2, dim=-1)
        alpha = self.get_alpha(sdf, normal, t_dirs, dists)[..., None]
        rgb = self.texture(feature, t_dirs, normal)
        weights = render_weight_from_alpha(alpha, ray_indices=ray_indices, n_rays=n_rays)
        opacity = accumulate_along_rays(weights, ray_indices, values=None, n_rays=n_rays)
        depth = accumulate_along_rays(weights, ray_indices, values=midpoints, n_rays=n_rays)
        comp_rgb = accumulate_along_rays(weights, ray_indices, values=rgb, n_rays=n_rays)
        comp_normal = F.normalize(accumulate_along_rays(weights, ray_indices, values=normal, n_rays=n_rays), p=2, dim=-1)
        out = {"comp_rgb": comp_rgb, "comp_normal": comp_normal, "opacity": opacity, "depth": depth,
               "rays_valid": opacity > 0,
               "num_samples": torch.as_tensor([len(t_starts)], dtype=torch.int32, device=rays.device)}
        if self.training:
            out.update({"sdf_samples": sdf, "sdf_grad_samples": sdf_grad, "weights": weights.view(-1),
                        "points": midpoints.view(-1), "intervals": dists.view(-1), "ray_indices": ray_indices.view(-1)})
            if fd:
                out["sdf_laplace_samples"] = sdf_laplace
        if self.config["learned_background"]:
            out_bg = self.forward_bg_(rays)
        else:
            out_bg = {"comp_rgb": self.background_color[None, :].expand(*comp_rgb.shape),
                      "num_samples": torch.zeros_like(out["num_samples"]),
                      "rays_valid": torch.zeros_like(out["rays_valid"])}
        out_full = {"comp_rgb": out["comp_rgb"] + out_bg["comp_rgb"] * (1.0 - out["opacity"]),
                    "num_samples": out["num_samples"] + out_bg["num_samples"],
                    "rays_valid": out["rays_valid"] | out_bg["rays_valid"]}
        return {**out, **{k + "_bg": v for k, v in out_bg.items()}, **{k + "_full": v for k, v in out_full.items()}}

    def forward(self, rays):
        out = self.forward_(rays) if self.training else chunk_batch(self.forward_, self.config["ray_chunk"], True, rays)
        return {**out, "inv_s": self.variance.inv_s}

    def regularizations(self, out):
        """models/neus.py:307-311: geometry + texture regularizers (both empty in the reference: models/base.py:27-28)"""
        return {}

    def train(self, mode=True):
        self.randomized = mode and self.config["randomized"]
        return super().train(mode=mode)

    def eval(self):
        self.randomized = False
        return super().eval()


def make(config):
    return {"nerf": NeRFModel, "neus": NeuSModel}[config["name"]](config)
