"""Oracle restatements for the radiance / material branch.
TEST INFRASTRUCTURE ONLY (same rules as oracle/__init__.py).

  vanilla_frequency   models/network_utils.py:14-40                      pinned (golden freq_srgb.npz)
  sh_encode           tcnn.Encoding(otype=SphericalHarmonics), call sites models/network_utils.py:98-99,
                      models/texture.py:312,348.  tiny-cuda-nn is absent and unpinned upstream =>
                      PARITY UNPINNED; this is the published real-SH polynomial basis (degree <= 5),
                      inputs in [0,1] mapped to [-1,1], as Instant-NGP documents it.
  reflect / NoV       models/texture.py:295-297
  texture_stage0      models/texture.py:292-327 (VolumeMixedMipSplitOcc.forward, stage == 0)
  rgb_to_srgb         lib/pbr/utils/nvdiffrecmc_util.py:95-103                pinned (golden freq_srgb.npz)
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def vanilla_frequency(x, n_frequencies, mask=None, x_scale=1.0, x_offset=0.0):
    x = x * x_scale + x_offset
    out = []
    for k in range(n_frequencies):
        m = 1.0 if mask is None else mask[k]
        f = float(2 ** k)
        out += [torch.sin(f * x) * m, torch.cos(f * x) * m]
    return torch.cat(out, -1)


def sh_encode(d01, degree=5):
    """Real spherical harmonics up to ``degree`` bands (degree**2 outputs) of 2*d01-1."""
    x, y, z = (d01[..., 0] * 2 - 1), (d01[..., 1] * 2 - 1), (d01[..., 2] * 2 - 1)
    xy, xz, yz = x * y, x * z, y * z
    x2, y2, z2 = x * x, y * y, z * z
    x4, y4, z4 = x2 * x2, y2 * y2, z2 * z2
    o = [torch.full_like(x, 0.28209479177387814)]
    if degree > 1:
        o += [-0.48860251190291987 * y, 0.48860251190291987 * z, -0.48860251190291987 * x]
    if degree > 2:
        o += [1.0925484305920792 * xy, -1.0925484305920792 * yz,
              0.94617469575755997 * z2 - 0.31539156525251999, -1.0925484305920792 * xz,
              0.54627421529603959 * x2 - 0.54627421529603959 * y2]
    if degree > 3:
        o += [0.59004358992664352 * y * (-3.0 * x2 + y2), 2.8906114426405538 * xy * z,
              0.45704579946446572 * y * (1.0 - 5.0 * z2), 0.3731763325901154 * z * (5.0 * z2 - 3.0),
              0.45704579946446572 * x * (1.0 - 5.0 * z2), 1.4453057213202769 * z * (x2 - y2),
              0.59004358992664352 * x * (-x2 + 3.0 * y2)]
    if degree > 4:
        o += [2.5033429417967046 * xy * (x2 - y2), 1.7701307697799304 * yz * (-3.0 * x2 + y2),
              0.94617469575756008 * xy * (7.0 * z2 - 1.0), 0.66904654355728921 * yz * (3.0 - 7.0 * z2),
              -3.1735664074561294 * z2 + 3.7024941420321507 * z4 + 0.31735664074561293,
              0.66904654355728921 * xz * (3.0 - 7.0 * z2),
              0.47308734787878004 * (x2 - y2) * (7.0 * z2 - 1.0),
              1.7701307697799304 * xz * (-x2 + 3.0 * y2),
              -3.7550144126950569 * x2 * y2 + 0.62583573544917614 * x4 + 0.62583573544917614 * y4]
    return torch.stack(o, -1)


def reflect_dirs(dirs, normals):
    """wi = -d; wo = 2 (wi.n) n - wi; NoV = n.wi   (models/texture.py:295-297)."""
    wi = -dirs
    wo = torch.sum(wi * normals, -1, keepdim=True) * normals * 2 - wi
    nov = torch.sum(normals * wi, -1, keepdim=True)
    return wo, nov


def _oracle_linear(x, w, b):
    from . import linear
    return linear(x, w, b)


def relu_mlp(x, params):
    """VanillaMLP without sphere init / weight norm: ReLU hidden layers (network_utils.py:145-157)."""
    h = x
    for i, p in enumerate(params):
        h = _oracle_linear(h, p["w"], p["b"])      # fp32, or bf16 operands inside oracle.mlp_precision("bf16")
        if i < len(params) - 1:
            h = F.relu(h)
    return h


def texture_stage0(features, dirs, normals, positions, nets, n_frequencies=6, sh_degree=5):
    """-> [S,7] = [diff_rgb (3), spec_rgb (3), blend (1)]; nets = dict(albedo, roughness, metallic, env)
    of relu_mlp parameter lists; color_activation = sigmoid (configs/...yaml:125)."""
    wo, _ = reflect_dirs(dirs, normals)
    xyz = vanilla_frequency(positions, n_frequencies)
    inp = torch.cat([features, xyz], -1)
    albedo6 = relu_mlp(inp, nets["albedo"])
    metallic2 = relu_mlp(inp, nets["metallic"])
    wo_enc = sh_encode((wo + 1.0) / 2.0, sh_degree)
    spec = relu_mlp(torch.cat([features, wo_enc], -1), nets["env"])
    diff_rgb = torch.sigmoid(albedo6[..., :3])
    blend = torch.sigmoid(metallic2[..., :1])
    spec_rgb = blend * torch.sigmoid(spec)
    diff_rgb = (1 - blend) * diff_rgb
    return torch.cat([diff_rgb, spec_rgb, blend], -1)


def rgb_to_srgb(f):
    return torch.where(f <= 0.0031308, f * 12.92, torch.pow(torch.clamp(f, 0.0031308), 1.0 / 2.4) * 1.055 - 0.055)


def synthetic_fg_lut(res=256):
    """Deterministic smooth stand-in [1,res,res,2] for the reference's load/bsdf/bsdf_256_256.bin (not in its
    repository); used by the stage-1 golden fixture and its tests only.  [0, y(roughness), x(NoV), :]."""
    c = (torch.arange(res, dtype=torch.float64) + 0.5) / res
    v, u = torch.meshgrid(c, c, indexing="ij")
    a = 1.0 - 0.6 * v * (1.0 - u) - 0.2 * (1 - u) ** 3
    b = 0.25 * (1.0 - u) ** 2 * (1.0 - 0.5 * v) + 0.02 * torch.sin(7.0 * u + 3.0 * v)
    return torch.stack([a, b], -1)[None].float()


def texture_stage1(features, dirs, normals, positions, nets, fg_lut, eval_diffuse, eval_specular,
                   n_frequencies=6, sh_degree=5):
    """Stage-1 colours [S,24] (models/texture.py:292-345).  ``eval_diffuse(normals)`` and
    ``eval_specular(wo, roughness)`` are the emitter lookups; fg_lut [1,H,W,2]."""
    from . import gridsample as ogs
    wo, nov = reflect_dirs(dirs, normals)
    xyz = vanilla_frequency(positions, n_frequencies)
    inp = torch.cat([features, xyz], -1)
    albedo6 = torch.sigmoid(relu_mlp(inp, nets["albedo"]))
    roughness = torch.sigmoid(relu_mlp(inp, nets["roughness"]))
    metallic2 = torch.sigmoid(relu_mlp(inp, nets["metallic"]))
    wo_enc = sh_encode((wo + 1.0) / 2.0, sh_degree)
    spec = torch.sigmoid(relu_mlp(torch.cat([features, wo_enc], -1), nets["env"]))
    diff_rgb, albedo = albedo6[..., :3], albedo6[..., 3:]
    blend, metallic = metallic2[..., :1], metallic2[..., 1:]
    spec_rgb = blend * spec
    diff_rgb = (1 - blend) * diff_rgb
    diff_pbr = (1 - metallic) * albedo * eval_diffuse(normals)
    spec_albedo = 0.04 * (1 - metallic) + metallic * albedo
    spec_light = eval_specular(wo, roughness)
    uv = torch.cat([nov.clamp(0.0, 1.0), roughness.clamp(0.0, 1.0)], -1)
    grid = (uv * 2.0 - 1.0).reshape(1, -1, 1, 2)
    fg = ogs.grid_sample_2d(fg_lut.permute(0, 3, 1, 2).to(uv.dtype), grid, "border", False)[0, :, :, 0].t()
    spec_ref = spec_albedo * fg[:, 0:1] + fg[:, 1:2]
    return torch.cat([diff_rgb, spec_rgb, blend, diff_pbr, spec_ref * spec_light, spec_ref, spec_light, albedo,
                      metallic, roughness], -1)


def secondary_shading_pbr(features, dirs, normals, positions, nets, fg_lut, eval_diffuse, eval_specular,
                          n_frequencies=6):
    """models/texture.py:386-427 -> [S,3]: split-sum shading with the specular lobe looked up along ``dirs``."""
    from . import gridsample as ogs
    _, nov = reflect_dirs(dirs, normals)
    inp = torch.cat([features, vanilla_frequency(positions, n_frequencies)], -1)
    albedo = torch.sigmoid(relu_mlp(inp, nets["albedo"]))[..., 3:]
    roughness = torch.sigmoid(relu_mlp(inp, nets["roughness"]))
    metallic = torch.sigmoid(relu_mlp(inp, nets["metallic"]))[..., 1:]
    diff = (1 - metallic) * albedo * eval_diffuse(normals)
    spec_albedo = 0.04 * (1 - metallic) + metallic * albedo
    spec_light = eval_specular(dirs, roughness)
    uv = torch.cat([nov.clamp(0.0, 1.0), roughness.clamp(0.0, 1.0)], -1)
    grid = (uv * 2.0 - 1.0).reshape(1, -1, 1, 2)
    fg = ogs.grid_sample_2d(fg_lut.permute(0, 3, 1, 2).to(uv.dtype), grid, "border", False)[0, :, :, 0].t()
    return diff + (spec_albedo * fg[:, 0:1] + fg[:, 1:2]) * spec_light
