"""Host-side mirror of the reference's ``models/ray_utils.py`` (I0): pinhole directions and camera-to-world rays,
the latter through the HIP ray generator (``rsdf_gen_rays``).

  get_ray_directions   models/ray_utils.py:9-29   (pixel centres +0.5, OpenGL: x right, y up, camera looks along -z)
  get_rays             models/ray_utils.py:32-56
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import ops


def get_ray_directions(W, H, fx, fy, cx, cy, use_pixel_centers=True, device=None):
    pc = 0.5 if use_pixel_centers else 0.0
    i, j = torch.meshgrid(torch.arange(W, dtype=torch.float32, device=device) + pc,
                          torch.arange(H, dtype=torch.float32, device=device) + pc, indexing="xy")
    return torch.stack([(i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)], -1)   # (H, W, 3)


def get_rays(directions, c2w, keepdim=False, normalize=False):
    """directions (H,W,3) with c2w (3,4), or (N,3) with c2w (N,3,4)/(1,3,4) -> (rays_o, rays_d), flat [N,3].
    ``normalize=True`` returns unit directions (what the systems concatenate into ``rays``, split_occ.py:103)."""
    assert directions.shape[-1] == 3
    if directions.ndim == 3 and c2w.ndim == 2:
        H, W = directions.shape[:2]
        yy, xx = torch.meshgrid(torch.arange(H, device=directions.device), torch.arange(W, device=directions.device),
                                indexing="ij")
        index = torch.zeros(1, dtype=torch.int64, device=directions.device)
        rays, _, _ = ops.gen_rays(index, yy.reshape(-1), xx.reshape(-1), directions, c2w[None])
    else:
        raise NotImplementedError("per-ray c2w: use ops.gen_rays with the view indices")
    rays_o, rays_d = rays[:, :3], rays[:, 3:]
    if not normalize:
        # gen_rays normalises; undo with the un-normalised length for callers that want raw directions
        d = directions.reshape(-1, 3)
        rays_d = rays_d * torch.linalg.norm(d @ c2w[:3, :3].T, dim=-1, keepdim=True)
    if keepdim:
        rays_o, rays_d = rays_o.reshape(*directions.shape), rays_d.reshape(*directions.shape)
    return rays_o, rays_d


def orbit_view_rays(W, H, seed=0, radius=4.0, fov=0.6911112, device=None):
    """One synthetic pinhole view on a sphere of ``radius`` looking at the origin (Blender / TensoIR
    ``camera_angle_x`` convention) -> rays [H*W, 6] = (origin, unit direction), generated on the device."""
    rng = np.random.default_rng(seed)
    az, el = rng.uniform(0, 2 * math.pi), rng.uniform(0.2, 1.0)
    eye = radius * np.array([math.cos(el) * math.cos(az), math.cos(el) * math.sin(az), math.sin(el)])
    fwd = -eye / np.linalg.norm(eye)
    right = np.cross(fwd, np.array([0.0, 0.0, 1.0]))
    right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    c2w = torch.tensor(np.stack([right, up, -fwd, eye], axis=1), dtype=torch.float32, device=device)  # OpenGL
    focal = 0.5 * W / math.tan(0.5 * fov)
    dirs = get_ray_directions(W, H, focal, focal, W / 2, H / 2, device=device)
    ro, rd = get_rays(dirs, c2w, normalize=True)
    return torch.cat([ro, rd], dim=-1)
