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


def seeded_param(name, shape, seed=0):
    """A parameter tensor that both sides of a fixture can regenerate from its NAME (order-independent): uniform in
    +-1/sqrt(fan_in) for matrices (nn.Linear's scale), +-0.1 for vectors.  tests/golden/make_golden.py fills the reference's
    128-wide radiance networks with these, the GPU tests fill the HIP mirrors: 0.9 MB of weights stay out of the fixture."""
    import zlib
    g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    bound = 1.0 / math.sqrt(shape[-1]) if len(shape) >= 2 else 0.1
    return (torch.rand(*shape, generator=g) * 2 - 1) * bound


def composite_like_cotangent(shape, seed=0):
    """A cotangent with the dynamic range per-sample gradients have in a render: randn rows scaled by weights spanning
    1e-8 ... 1 (composite weights of a pruned ray) with ~10 % exact zeros (samples behind a saturated surface)."""
    g = torch.Generator().manual_seed(seed)
    w = 10.0 ** (-8.0 * torch.rand(shape[0], generator=g))
    w[torch.rand(shape[0], generator=g) < 0.1] = 0.0
    w[0] = 1.0
    return torch.randn(*shape, generator=g) * w[:, None]


def late_regime_field(hidden=64, seed=0, n_levels=16, log2_T=19, base=32, scale=1.447269237440378, table_amp=3e-2, feat=48):
    """Oracle-side parameters of the late-training c1 field of tests/test_gpu_late_regime.py (the same recipe on the CPU): a
    table of +-3e-2, un-zeroed hash columns of the first layer, the output bias shifted so that central rays cross sdf = 0."""
    meta, n_params = oracle.grid_meta(n_levels, 2, log2_T, base, scale)
    g = torch.Generator().manual_seed(seed)
    table = ((torch.rand(n_params, generator=g) * 2 - 1) * table_amp).requires_grad_(True)
    mlp = oracle.sphere_init_mlp_params(3 + 2 * n_levels, feat, hidden, 2, seed=seed + 1)
    with torch.no_grad():
        mlp[0]["v"][:, 3:] = torch.randn(mlp[0]["v"][:, 3:].shape, generator=g) * 0.3
        mlp[-1]["b"][0] += 0.25
    for p in mlp:
        for k in p:
            p[k] = p[k].detach().clone().requires_grad_(True)
    return meta, table, mlp


def oracle_gradient_sensitivity(render, leaves, sdf7, cot, trials=3, seed=0):
    """How far the ORACLE's own parameter gradients move when its stencil inputs move by one fp32 ulp (VERDICT r05 item 1:
    the measured conditioning of the reference's backward, instead of a formula for it).  ``render(sdf7_given)`` -> dict of
    outputs with a graph to ``leaves`` (name -> leaf tensor); ``cot`` = {output name: cotangent}.  Every trial replaces each
    of the 7 S stencil values by its fp32 neighbour above or below (random signs) and takes the backward again; returns
    (base gradients, {name: largest |gradient movement| over the trials / largest |base gradient| of that tensor})."""
    names = list(leaves)

    def grads(s7):
        for t in leaves.values():
            t.grad = None
        out = render(s7)
        sum((out[k] * c).sum() for k, c in cot.items()).backward()
        return {n: (leaves[n].grad.detach().clone() if leaves[n].grad is not None else torch.zeros_like(leaves[n]))
                for n in names}                         # (a leaf the outputs do not depend on: zeros)

    base = grads(sdf7)
    g = torch.Generator().manual_seed(seed)
    moved = {n: 0.0 for n in names}
    for _ in range(trials):
        up = torch.rand(sdf7.shape, generator=g) < 0.5
        s7 = torch.where(up, torch.nextafter(sdf7, torch.full_like(sdf7, float("inf"))),
                         torch.nextafter(sdf7, torch.full_like(sdf7, float("-inf"))))
        got = grads(s7)
        for n in names:
            moved[n] = max(moved[n], float((got[n] - base[n]).abs().max()) / (float(base[n].abs().max()) + 1e-30))
    return base, moved
