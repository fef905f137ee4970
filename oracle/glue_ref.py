"""Oracle restatement of the reference-OWNED glue of the hot path.  TEST INFRASTRUCTURE ONLY.

Unlike the tcnn/nerfacc arithmetic, these functions ARE pinned: ``tests/gen_golden.py`` runs the reference's own
``models/*.py`` (imported unchanged from /root/reference) and ``tests/test_golden_glue.py`` checks this file
against the committed outputs.  Each function cites the reference lines it follows.
"""
import torch
import torch.nn.functional as F

from . import nerfacc_ref as N
from . import tcnn_ref as T


def scale_anything(dat, inp_scale, tgt_scale):
    """reference models/utils.py:108-113"""
    dat = (dat - inp_scale[0]) / (inp_scale[1] - inp_scale[0])
    return dat * (tgt_scale[1] - tgt_scale[0]) + tgt_scale[0]


def contract_to_unisphere(x, radius, contraction_type):
    """reference models/geometry.py:17-29"""
    x = scale_anything(x, (-radius, radius), (0, 1))
    if contraction_type == N.ContractionType.AABB:
        return x
    if contraction_type == N.ContractionType.UN_BOUNDED_SPHERE:
        x = x * 2 - 1
        mag = x.norm(dim=-1, keepdim=True)
        x = torch.where(mag > 1, (2 - 1 / mag) * (x / mag), x)
        return x / 4 + 0.5
    raise NotImplementedError


class _TruncExp(torch.autograd.Function):
    """reference models/utils.py:53-68: exp forward, gradient uses exp(clamp(x, max=15))"""

    @staticmethod
    def forward(ctx, x):
        x = x.float()
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        return g * torch.exp(torch.clamp(ctx.saved_tensors[0], max=15))


trunc_exp = _TruncExp.apply


def neus_alpha(sdf, normal, dirs, dists, inv_s, cos_anneal_ratio):
    """reference models/neus.py:117-139 (inv_s: 0-dim tensor exp(10*variance))"""
    inv_s = inv_s.reshape(1, 1).clip(1e-6, 1e6).expand(sdf.shape[0], 1)
    true_cos = (dirs * normal).sum(-1, keepdim=True)
    iter_cos = -(F.relu(-true_cos * 0.5 + 0.5) * (1.0 - cos_anneal_ratio) + F.relu(-true_cos) * cos_anneal_ratio)
    est_next = sdf[..., None] + iter_cos * dists.reshape(-1, 1) * 0.5
    est_prev = sdf[..., None] - iter_cos * dists.reshape(-1, 1) * 0.5
    prev_cdf, next_cdf = torch.sigmoid(est_prev * inv_s), torch.sigmoid(est_next * inv_s)
    p, c = prev_cdf - next_cdf, prev_cdf
    return ((p + 1e-5) / (c + 1e-5)).view(-1).clip(0.0, 1.0)


def volume_density(points, enc_with_net, radius, contraction_type, density_bias=-1.0):
    """reference models/geometry.py:122-130 (density_activation trunc_exp): -> density[n], feature[n, C]"""
    x = contract_to_unisphere(points, radius, contraction_type)
    out = enc_with_net(x.view(-1, 3)).view(*x.shape[:-1], -1).float()
    return trunc_exp(out[..., 0] + float(density_bias)), out


def volume_radiance(features, dirs, sh_encoding, network, color_activation=None, *args):
    """reference models/texture.py:23-30: [feature | SH((d+1)/2) | extra] -> MLP -> rgb"""
    d = (dirs + 1.0) / 2.0
    inp = torch.cat([features.view(-1, features.shape[-1]), sh_encoding(d.view(-1, 3))] +
                    [a.view(-1, a.shape[-1]) for a in args], dim=-1)
    color = network(inp).view(*features.shape[:-1], 3).float()
    if color_activation == "sigmoid":
        color = torch.sigmoid(color)
    return color


def nerf_forward(rays, enc_with_net, sh_encoding, color_net, grid, scene_aabb, radius, render_step_size,
                 background_color, stratified=False):
    """reference models/nerf.py:61-127 (learned_background=False branch): march -> fields -> composite"""
    n_rays = rays.shape[0]
    rays_o, rays_d = rays[:, 0:3], rays[:, 3:6]
    ctype = N.ContractionType.AABB

    def sigma_fn(t_starts, t_ends, ray_indices):
        pos = rays_o[ray_indices.long()] + rays_d[ray_indices.long()] * (t_starts + t_ends) / 2.0
        return volume_density(pos, enc_with_net, radius, ctype)[0][..., None]

    with torch.no_grad():
        ray_indices, t_starts, t_ends = N.ray_marching(
            rays_o, rays_d, scene_aabb=scene_aabb, grid=grid, sigma_fn=sigma_fn, near_plane=None, far_plane=None,
            render_step_size=render_step_size, stratified=stratified, cone_angle=0.0, alpha_thre=0.0)
    ray_indices = ray_indices.long()
    t_dirs = rays_d[ray_indices]
    midpoints = (t_starts + t_ends) / 2.0
    positions = rays_o[ray_indices] + t_dirs * midpoints
    density, feature = volume_density(positions, enc_with_net, radius, ctype)
    rgb = volume_radiance(feature, t_dirs, sh_encoding, color_net)
    weights = N.render_weight_from_density(t_starts, t_ends, density[..., None], ray_indices=ray_indices, n_rays=n_rays)
    opacity = N.accumulate_along_rays(weights, ray_indices, values=None, n_rays=n_rays)
    depth = N.accumulate_along_rays(weights, ray_indices, values=midpoints, n_rays=n_rays)
    comp_rgb = N.accumulate_along_rays(weights, ray_indices, values=rgb, n_rays=n_rays)
    comp_rgb = comp_rgb + background_color * (1.0 - opacity)
    return dict(comp_rgb=comp_rgb, opacity=opacity, depth=depth, rays_valid=opacity > 0,
                num_samples=torch.as_tensor([len(t_starts)], dtype=torch.int32), weights=weights.view(-1),
                points=midpoints.view(-1), intervals=(t_ends - t_starts).view(-1), ray_indices=ray_indices.view(-1))


def sdf_with_analytic_grad(points, encoding, sdf_mlp, radius, include_xyz=True):
    """reference models/geometry.py:158-180 (grad_type analytic, AABB): sdf, grad (create_graph), feature"""
    points = points.requires_grad_(True)
    x = contract_to_unisphere(points, radius, N.ContractionType.AABB)
    enc = encoding(x.view(-1, 3))
    inp = torch.cat([x * 2.0 - 1.0, enc], dim=-1) if include_xyz else enc
    out = sdf_mlp(inp).float()
    sdf = out[..., 0]
    (grad,) = torch.autograd.grad(sdf, points, grad_outputs=torch.ones_like(sdf), create_graph=True,
                                  retain_graph=True, only_inputs=True)
    return sdf, grad, out


def sdf_with_finite_difference(points, encoding, sdf_mlp, radius, eps, level_mask=None, include_xyz=True):
    """reference models/geometry.py:181-199 (grad_type finite_difference): six +-eps taps clamped to the box, PLAIN AABB
    scaling of the taps (:194), central differences and the 7-point laplace; ``level_mask`` is the 0/1 column mask of
    ProgressiveBandHashGrid (models/network_utils.py:56-58).  -> sdf, grad, feature, laplace"""
    def field(x01):
        enc = encoding(x01.view(-1, 3))
        if level_mask is not None:
            enc = enc * level_mask
        inp = torch.cat([x01.view(-1, 3) * 2.0 - 1.0, enc], dim=-1) if include_xyz else enc
        return sdf_mlp(inp).float()
    x = contract_to_unisphere(points, radius, N.ContractionType.AABB)
    out = field(x)
    sdf = out[..., 0]
    offsets = torch.as_tensor([[eps, 0.0, 0.0], [-eps, 0.0, 0.0], [0.0, eps, 0.0], [0.0, -eps, 0.0],
                               [0.0, 0.0, eps], [0.0, 0.0, -eps]]).to(points)
    taps = (points[..., None, :] + offsets).clamp(-radius, radius)
    taps = scale_anything(taps, (-radius, radius), (0, 1))
    tap_sdf = field(taps)[..., 0].view(*points.shape[:-1], 6)
    grad = 0.5 * (tap_sdf[..., 0::2] - tap_sdf[..., 1::2]) / eps
    laplace = (tap_sdf[..., 0::2] + tap_sdf[..., 1::2] - 2 * sdf[..., None]).sum(-1) / (eps ** 2)
    return sdf, grad, out, laplace


def neus_forward(rays, encoding, sdf_mlp, sh_encoding, color_net, inv_s, grid, scene_aabb, radius, render_step_size,
                 cos_anneal_ratio, background_color, fd_eps=None, level_mask=None):
    """reference models/neus.py:205-287 (no learned background); analytic gradients, or finite differences + laplace
    when ``fd_eps`` is given"""
    n_rays = rays.shape[0]
    rays_o, rays_d = rays[:, 0:3], rays[:, 3:6]
    with torch.no_grad():
        ray_indices, t_starts, t_ends = N.ray_marching(
            rays_o, rays_d, scene_aabb=scene_aabb, grid=grid, alpha_fn=None, near_plane=None, far_plane=None,
            render_step_size=render_step_size, stratified=False, cone_angle=0.0, alpha_thre=0.0)
    ray_indices = ray_indices.long()
    t_dirs = rays_d[ray_indices]
    midpoints = (t_starts + t_ends) / 2.0
    positions = rays_o[ray_indices] + t_dirs * midpoints
    dists = t_ends - t_starts
    laplace = None
    if fd_eps is None:
        sdf, sdf_grad, feature = sdf_with_analytic_grad(positions, encoding, sdf_mlp, radius)
    else:
        sdf, sdf_grad, feature, laplace = sdf_with_finite_difference(positions, encoding, sdf_mlp, radius, fd_eps,
                                                                     level_mask)
    normal = F.normalize(sdf_grad, p=2, dim=-1)
    alpha = neus_alpha(sdf, normal, t_dirs, dists, inv_s, cos_anneal_ratio)[..., None]
    rgb = volume_radiance(feature, t_dirs, sh_encoding, color_net, "sigmoid", normal)
    weights = N.render_weight_from_alpha(alpha, ray_indices=ray_indices, n_rays=n_rays)
    opacity = N.accumulate_along_rays(weights, ray_indices, values=None, n_rays=n_rays)
    depth = N.accumulate_along_rays(weights, ray_indices, values=midpoints, n_rays=n_rays)
    comp_rgb = N.accumulate_along_rays(weights, ray_indices, values=rgb, n_rays=n_rays)
    comp_normal = F.normalize(N.accumulate_along_rays(weights, ray_indices, values=normal, n_rays=n_rays), p=2, dim=-1)
    comp_rgb_full = comp_rgb + background_color[None, :].expand(*comp_rgb.shape) * (1.0 - opacity)
    out = dict(comp_rgb=comp_rgb, comp_normal=comp_normal, opacity=opacity, depth=depth, rays_valid=opacity > 0,
               num_samples=torch.as_tensor([len(t_starts)], dtype=torch.int32), sdf_samples=sdf,
               sdf_grad_samples=sdf_grad, weights=weights.view(-1), points=midpoints.view(-1),
               intervals=dists.view(-1), ray_indices=ray_indices.view(-1), comp_rgb_full=comp_rgb_full,
               rays_valid_full=opacity > 0)
    if laplace is not None:
        out["sdf_laplace_samples"] = laplace
    return out
