"""A small analytic scene as a resident synthetic dataset (the reference's datasets are I/O and out of scope, and
toaster_disney is not in this image): a sphere and an axis-aligned box with a smooth, known albedo under a constant
white environment, ray-traced in closed form.  It gives the training step something to converge ON: the config[3]
workload of bench.py, the 2-rank step test and the convergence proxy of tests/test_gpu_c4.py (HIP model and CPU oracle
trained side by side on the same images).

Layout follows datasets/tensoir_synthetic.py:156-160 (everything resident on one device): ``all_images`` [V,H,W,3]
(sRGB, as the reference's PNG-derived images), ``all_fg_masks`` [V,H,W], ``directions`` [H,W,3] (OpenGL pinhole, pixel
centres), ``all_c2w`` [V,3,4].
"""
from __future__ import annotations

import math

import numpy as np
import torch


def _srgb(f):
    """lib/pbr/utils/nvdiffrecmc_util.py:95-103."""
    return torch.where(f <= 0.0031308, f * 12.92, torch.pow(torch.clamp(f, min=0.0031308), 1.0 / 2.4) * 1.055 - 0.055)


def albedo_at(p):
    """Smooth albedo in [0.15, 0.85]^3 as a function of position."""
    return 0.5 + 0.35 * torch.stack([torch.sin(4.0 * p[..., 0] + 0.3), torch.sin(3.0 * p[..., 1] - 0.7),
                                     torch.cos(5.0 * p[..., 2])], -1)


def trace(ro, rd, sphere=(0.0, 0.0, 0.05, 0.42), box=((0.25, -0.45, -0.4), (0.6, -0.05, 0.1))):
    """Nearest hit of rays with the sphere (cx, cy, cz, r) and the box (min, max) -> (hit bool [N], t [N])."""
    c = torch.tensor(sphere[:3], dtype=ro.dtype)
    oc = ro - c
    b = (oc * rd).sum(-1)
    disc = b * b - ((oc * oc).sum(-1) - sphere[3] ** 2)
    ts = torch.where(disc >= 0, -b - torch.sqrt(disc.clamp(min=0)), torch.full_like(b, float("inf")))
    ts = torch.where(ts > 0, ts, torch.full_like(ts, float("inf")))
    lo, hi = torch.tensor(box[0], dtype=ro.dtype), torch.tensor(box[1], dtype=ro.dtype)
    inv = 1.0 / torch.where(rd.abs() < 1e-12, torch.full_like(rd, 1e-12), rd)
    t0, t1 = (lo - ro) * inv, (hi - ro) * inv
    tn, tf = torch.minimum(t0, t1).amax(-1), torch.maximum(t0, t1).amin(-1)
    tb = torch.where((tn <= tf) & (tn > 0), tn, torch.full_like(tn, float("inf")))
    t = torch.minimum(ts, tb)
    return torch.isfinite(t), t


def orbit_c2w(n_views, radius=4.0, seed=0):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_views):
        az, el = rng.uniform(0, 2 * math.pi), rng.uniform(0.15, 1.1)
        eye = radius * np.array([math.cos(el) * math.cos(az), math.cos(el) * math.sin(az), math.sin(el)])
        fwd = -eye / np.linalg.norm(eye)
        right = np.cross(fwd, np.array([0.0, 0.0, 1.0]))
        right /= np.linalg.norm(right)
        up = np.cross(right, fwd)
        out.append(np.stack([right, up, -fwd, eye], axis=1))       # OpenGL camera-to-world
    return torch.tensor(np.stack(out), dtype=torch.float32)


def make_dataset(n_views=8, W=128, H=128, fov=0.6911112, seed=0, device=None):
    """-> dict(all_images, all_fg_masks, directions, all_c2w, w, h) generated on the CPU in fp32 (identical for every
    consumer), then moved to ``device``."""
    focal = 0.5 * W / math.tan(0.5 * fov)
    i, j = torch.meshgrid(torch.arange(W, dtype=torch.float32) + 0.5, torch.arange(H, dtype=torch.float32) + 0.5,
                          indexing="xy")
    directions = torch.stack([(i - W / 2) / focal, -(j - H / 2) / focal, -torch.ones_like(i)], -1)   # ray_utils.py:9-29
    c2w = orbit_c2w(n_views, seed=seed)
    images, masks = [], []
    for v in range(n_views):
        rd = directions.reshape(-1, 3) @ c2w[v, :3, :3].T
        rd = rd / rd.norm(dim=-1, keepdim=True)
        ro = c2w[v, :3, 3].expand_as(rd)
        hit, t = trace(ro, rd)
        p = ro + torch.where(hit, t, torch.zeros_like(t))[:, None] * rd
        lin = torch.where(hit[:, None], albedo_at(p), torch.ones_like(p))
        images.append(_srgb(lin).reshape(H, W, 3))
        masks.append(hit.float().reshape(H, W))
    ds = {"all_images": torch.stack(images), "all_fg_masks": torch.stack(masks), "directions": directions,
          "all_c2w": c2w, "w": W, "h": H}
    if device is not None:
        ds = {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in ds.items()}
    return ds
