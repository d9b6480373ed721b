"""Oracle for the analytic-normal rows (SURVEY.md 8a H4 analytic variant, H5 curvature).  TEST INFRASTRUCTURE ONLY.

  hashgrid_encode_t   the oracle's hash grid (oracle/risesdf_oracle.c orc_hashgrid_fwd; tcnn semantics, PARITY
                      UNPINNED like H1) written with torch ops so that autograd differentiates it to any order
                      in x and the table: indices from the bit-exact C routine, trilinear weights in torch.
  volume_sdf_analytic models/geometry.py:206-228 (grad = autograd.grad(sdf, points, create_graph=True))
  curvature           models/geometry.py:246-282 (PermutoSDF curvature term); the random directions are an input
                      (the reference draws torch.rand_like on the device stream)
Validated against the C forward in tests/test_oracle_analytic.py; restates reference code that needs tcnn's
double-backward, which cannot be run here.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import contract_aabb, hashgrid_indices, vanilla_mlp


def hashgrid_encode_t(x, table, meta, n_active_levels=None):
    """x [S,3] in [0,1] (any float dtype, may require grad), table [n_params] -> [S, L*F]."""
    L, Fd = meta.n_levels, meta.n_features
    idx = hashgrid_indices(x.detach(), meta).long()                      # [S,L,8], offsets included
    tab = table.reshape(-1, Fd)
    outs = []
    for l in range(L):
        if n_active_levels is not None and l >= n_active_levels:
            outs.append(torch.zeros(x.shape[0], Fd, dtype=x.dtype))
            continue
        # position in fp32 exactly as the kernel (fmaf(scale, x, 0.5)), cell from the fp32 value; the
        # fractional weight carries the derivative d pos / d x = scale
        scale = float(meta.scale[l])
        pos = x * scale + 0.5
        cell = torch.floor(torch.from_numpy(
            (np.float64(np.float32(scale)) * x.detach().to(torch.float32).double().numpy() + 0.5)
            .astype(np.float32)).double()).to(x.dtype)
        w = pos - cell
        acc = 0
        for c in range(8):
            wc = 1
            for d in range(3):
                wc = wc * (w[:, d] if (c >> d) & 1 else 1 - w[:, d])
            acc = acc + wc[:, None] * tab[idx[:, l, c]]
        outs.append(acc)
    return torch.cat(outs, -1)


def field(points, table, meta, mlp_params, radius, n_active_levels=None):
    """points [S,3] world -> MLP output [S,D] (composite encoding with xyz, models/network_utils.py:78-79)."""
    xu = contract_aabb(points, radius)
    enc = hashgrid_encode_t(xu, table, meta, n_active_levels)
    h = torch.cat([xu * 2.0 - 1.0, enc], -1)
    return _mlp(h, mlp_params)


def _mlp(h, params):
    from . import weight_norm_effective
    for i, p in enumerate(params):
        w = weight_norm_effective(p["g"], p["v"]) if "g" in p else p["w"]
        h = F.linear(h, w.to(h.dtype), p["b"].to(h.dtype))
        if i < len(params) - 1:
            h = F.softplus(h, beta=100)
    return h


def volume_sdf_analytic(points, table, meta, mlp_params, *, radius, n_active_levels=None):
    """-> (sdf [S], grad [S,3] with graph, feature [S,D])  (geometry.py:209-228)."""
    if not points.requires_grad:
        points = points.clone().requires_grad_(True)
    out = field(points, table, meta, mlp_params, radius, n_active_levels)
    sdf = out[:, 0]
    (grad,) = torch.autograd.grad(sdf, points, torch.ones_like(sdf), create_graph=True)
    return sdf, grad, out


def curvature(points, grad, rand_directions, table, meta, mlp_params, *, radius, n_active_levels=None):
    """geometry.py:246-282: angle / pi between the normal at x and the analytic normal at x + 1e-4 * tangent."""
    eps = 1e-4
    rd = F.normalize(rand_directions, dim=-1, eps=1e-6)
    normal = F.normalize(grad, dim=-1, eps=1e-6)
    tangent = torch.cross(normal, rd, dim=-1)
    pd = points + eps * tangent
    sdf_d = field(pd, table, meta, mlp_params, radius, n_active_levels)[:, 0]
    (grad_d,) = torch.autograd.grad(sdf_d, pd, torch.ones_like(sdf_d), create_graph=True)
    dot = torch.sum(F.normalize(grad, dim=-1, eps=1e-6) * F.normalize(grad_d, dim=-1, eps=1e-6), dim=-1)
    return torch.acos(torch.clamp(dot, -1.0 + 1e-6, 1.0 - 1e-6)) / np.pi
