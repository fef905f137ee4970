"""Procedural stand-in for nerf-synthetic/lego (no dataset is on disk and there is no network).

Same tensors, shapes and camera model as the reference's ``datasets/blender.py:24-84``: ``all_images[N,H,W,3]``,
``all_fg_masks[N,H,W]``, ``all_c2w[N,3,4]``, ``directions[H,W,3]`` (pixel centres, focal from ``camera_angle_x``),
cameras on a sphere of radius 4.03 looking at the origin.  The object is a CSG-like union of boxes and spheres
inside radius 1.0 rendered analytically (closest ray/primitive hit, lambert shading), so images are exact and cheap.
"""
import math

import torch


def get_ray_directions(W, H, fx, fy, cx, cy):
    """reference models/ray_utils.py:9-20 (pixel centres, OpenGL camera: -z forward, y up)"""
    i, j = torch.meshgrid(torch.arange(W, dtype=torch.float32) + 0.5, torch.arange(H, dtype=torch.float32) + 0.5,
                          indexing="xy")
    return torch.stack([(i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)], -1)


def get_rays(directions, c2w):
    """reference models/ray_utils.py:23-43 for directions [N,3] and c2w [N,3,4]"""
    rays_d = (directions[:, None, :] * c2w[:, :3, :3]).sum(-1)
    rays_o = c2w[:, :, 3].expand(rays_d.shape)
    return rays_o, rays_d


def _look_at_poses(n, radius, generator):
    """camera-to-world [n,3,4] on the upper hemisphere, looking at the origin (blender convention)"""
    u = torch.rand(n, generator=generator)
    phi = torch.rand(n, generator=generator) * 2 * math.pi
    z = 0.15 + 0.8 * u
    r = torch.sqrt(1 - z * z)
    pos = torch.stack([r * torch.cos(phi), r * torch.sin(phi), z], -1) * radius
    fwd = -pos / pos.norm(dim=-1, keepdim=True)             # camera looks along -z_cam
    up = torch.tensor([0.0, 0.0, 1.0]).expand_as(fwd)
    right = torch.cross(fwd, up, dim=-1)
    right = right / right.norm(dim=-1, keepdim=True)
    true_up = torch.cross(right, fwd, dim=-1)
    rot = torch.stack([right, true_up, -fwd], dim=-1)        # columns: x_cam, y_cam, z_cam in world
    return torch.cat([rot, pos[..., None]], dim=-1)


# (centre, half-size, colour) boxes and (centre, radius, colour) spheres
BOXES = [((0.0, 0.0, -0.35), (0.9, 0.55, 0.12), (0.85, 0.75, 0.2)), ((-0.45, 0.0, 0.05), (0.3, 0.4, 0.28), (0.8, 0.2, 0.15)),
         ((0.45, 0.1, 0.0), (0.22, 0.22, 0.45), (0.2, 0.4, 0.8)), ((0.1, -0.35, 0.25), (0.5, 0.08, 0.08), (0.3, 0.3, 0.3))]
SPHERES = [((0.45, 0.1, 0.6), 0.25, (0.9, 0.9, 0.9)), ((-0.6, -0.45, -0.1), 0.2, (0.2, 0.7, 0.3)),
           ((-0.6, 0.45, -0.1), 0.2, (0.2, 0.7, 0.3))]


def render_analytic(rays_o, rays_d):
    """-> rgb[n,3], mask[n]: closest hit among the primitives, lambert + ambient shading"""
    n = rays_o.shape[0]
    dev = rays_o.device
    t_best = torch.full((n,), 1e10, device=dev)
    col = torch.zeros(n, 3, device=dev)
    nrm = torch.zeros(n, 3, device=dev)
    for c, h, rgb in BOXES:
        c, h = torch.tensor(c, device=dev), torch.tensor(h, device=dev)
        inv = 1.0 / rays_d
        t1, t2 = (c - h - rays_o) * inv, (c + h - rays_o) * inv
        tn, tf = torch.minimum(t1, t2).amax(-1), torch.maximum(t1, t2).amin(-1)
        hit = (tn < tf) & (tn > 0) & (tn < t_best)
        p = rays_o + tn[:, None] * rays_d
        q = (p - c) / h
        ax = q.abs().argmax(-1)
        nn_ = torch.zeros(n, 3, device=dev).scatter_(1, ax[:, None], torch.sign(q.gather(1, ax[:, None])))
        t_best = torch.where(hit, tn, t_best)
        col = torch.where(hit[:, None], torch.tensor(rgb, device=dev).expand(n, 3), col)
        nrm = torch.where(hit[:, None], nn_, nrm)
    for c, r, rgb in SPHERES:
        c = torch.tensor(c, device=dev)
        oc = rays_o - c
        b = (oc * rays_d).sum(-1)
        disc = b * b - ((oc * oc).sum(-1) - r * r)
        tn = -b - torch.sqrt(disc.clamp_min(0))
        hit = (disc > 0) & (tn > 0) & (tn < t_best)
        p = rays_o + tn[:, None] * rays_d
        t_best = torch.where(hit, tn, t_best)
        col = torch.where(hit[:, None], torch.tensor(rgb, device=dev).expand(n, 3), col)
        nrm = torch.where(hit[:, None], (p - c) / r, nrm)
    mask = t_best < 1e9
    light = torch.nn.functional.normalize(torch.tensor([0.4, -0.3, 0.85], device=dev), dim=0)
    shade = 0.35 + 0.65 * (nrm * light).sum(-1).clamp_min(0)
    return col * shade[:, None] * mask[:, None], mask.float()


class SyntheticBlender:
    """tensors named as in the reference's BlenderDatasetBase (datasets/blender.py:66-71)"""

    def __init__(self, n_images=100, w=800, h=800, device="cuda", seed=0, camera_angle_x=0.6911112070083618,
                 environment=False):
        """``environment``: unmasked captures (configs/neus-dtu.yaml: apply_mask false) -- pixels that miss the object
        show a smooth direction-dependent backdrop, the target of the learned NeRF++ background"""
        g = torch.Generator().manual_seed(seed)
        self.w, self.h = w, h
        focal = 0.5 * w / math.tan(0.5 * camera_angle_x)
        self.directions = get_ray_directions(w, h, focal, focal, w // 2, h // 2).to(device)
        self.all_c2w = _look_at_poses(n_images, 4.031128874, g).float().to(device)
        imgs, masks = [], []
        dirs = self.directions.view(-1, 3)
        for i in range(n_images):
            c2w = self.all_c2w[i]
            rd = torch.nn.functional.normalize((dirs[:, None, :] * c2w[None, :3, :3]).sum(-1), dim=-1)
            ro = c2w[:, 3].expand_as(rd)
            rgb, m = render_analytic(ro, rd)
            if environment:
                sky = 0.5 + 0.5 * torch.stack([torch.sin(3.0 * rd[:, 0] + 0.5), torch.sin(2.0 * rd[:, 1] + 1.5),
                                               torch.sin(4.0 * rd[:, 2] + 2.5)], -1)
                rgb = rgb + sky * (1 - m[:, None])
            imgs.append(rgb.view(h, w, 3))
            masks.append(m.view(h, w))
        self.all_images = torch.stack(imgs).float()
        self.all_fg_masks = torch.stack(masks).float()
        self.apply_mask = not environment  # datasets/blender.py:37 / datasets/dtu.py with apply_mask: false

    def sample_rays(self, n_rays, generator=None, background="random"):
        """training ray batch (reference systems/nerf.py:33-85, batch_image_sampling=True)"""
        dev = self.all_images.device
        index = torch.randint(0, len(self.all_images), (n_rays,), device=dev, generator=generator)
        x = torch.randint(0, self.w, (n_rays,), device=dev, generator=generator)
        y = torch.randint(0, self.h, (n_rays,), device=dev, generator=generator)
        rays_o, rays_d = get_rays(self.directions[y, x], self.all_c2w[index])
        rgb = self.all_images[index, y, x].view(-1, 3)
        fg = self.all_fg_masks[index, y, x].view(-1)
        rays = torch.cat([rays_o, torch.nn.functional.normalize(rays_d, p=2, dim=-1)], dim=-1)
        if background == "random":
            bg = torch.rand(3, device=dev, generator=generator)
        else:
            bg = torch.ones(3, device=dev)
        if self.apply_mask:
            rgb = rgb * fg[..., None] + bg * (1 - fg[..., None])
        return rays, rgb, fg, bg
