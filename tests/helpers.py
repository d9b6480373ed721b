"""Shared synthetic inputs for parity tests (seeded, explicit random tensors: SURVEY.md 7 item 9)."""
import math

import numpy as np
import torch

import oracle


def camera_rays(W, H, seed=0, radius=4.0, fov=0.6911112):
    """One pinhole view on a sphere of radius 4 looking at the origin."""
    rng = np.random.default_rng(seed)
    az, el = rng.uniform(0, 2 * math.pi), rng.uniform(0.2, 1.0)
    eye = radius * np.array([math.cos(el) * math.cos(az), math.cos(el) * math.sin(az), math.sin(el)])
    fwd = -eye / np.linalg.norm(eye)
    right = np.cross(fwd, np.array([0.0, 0.0, 1.0]))
    right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    c2w = torch.tensor(np.stack([right, up, -fwd, eye], axis=1), dtype=torch.float32)  # OpenGL
    focal = 0.5 * W / math.tan(0.5 * fov)
    dirs = oracle.get_ray_directions(W, H, focal, focal, W / 2, H / 2)
    ro, rd = oracle.get_rays(dirs, c2w)
    return oracle.make_rays(ro, rd)


def sphere_binary(res=32, r_in=0.3, r_out=0.8, radius=1.5):
    """Occupancy shell around a sphere, bool [res,res,res]."""
    c = (torch.arange(res, dtype=torch.float32) + 0.5) / res * 2 * radius - radius
    x, y, z = torch.meshgrid(c, c, c, indexing="ij")
    d = torch.sqrt(x * x + y * y + z * z)
    return (d > r_in) & (d < r_out)


def small_field(seed=0, n_levels=4, base=16, log2_T=14, hidden=32, feat=13, table_scale=1e-2):
    """A small hash grid + sphere-initialised SDF MLP (oracle-side parameters)."""
    meta, n_params = oracle.grid_meta(n_levels, 2, log2_T, base, 1.5)
    g = torch.Generator().manual_seed(seed)
    table = (torch.rand(n_params, generator=g) * 2 - 1) * table_scale
    mlp = oracle.sphere_init_mlp_params(3 + 2 * n_levels, feat, hidden, 2, seed=seed + 1)
    # make weight_norm non-trivial
    for p in mlp:
        p["g"] = p["g"] * (1 + 0.05 * torch.randn(p["g"].shape, generator=g))
        p["b"] = p["b"] + 0.01 * torch.randn(p["b"].shape, generator=g)
    return meta, table, mlp, dict(n_levels=n_levels, base=base, log2_T=log2_T, hidden=hidden,
                                  feat=feat)


def rel_err(a, b, eps=1e-12):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + eps))
