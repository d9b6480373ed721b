"""CPU: the oracle's environment-light / stage-1 restatements -- properties, and the stage-1 golden vectors
made by the reference's own light.py + texture.py code (tests/golden/make_golden.py)."""
import os

import numpy as np
import torch

from oracle import envlight as E
from oracle import texture as otex
from test_oracle_texture import load, nets_from


def test_prefilters_preserve_a_constant_environment():
    c = torch.full((6, 16, 16, 3), 0.5, dtype=torch.float64)
    # cosine-lobe integral with the reference's constants: sum area * clamp(cos) / 3.141592 ~ 1.08 (:110-139)
    d = E.diffuse_cubemap(c)
    assert float((d / 0.5 - d[0, 0, 0, 0] / 0.5).abs().max()) < 5e-2 and 1.0 < float(d.mean() / 0.5) < 1.15
    for r in (0.08, 0.3, 1.0):
        s = E.specular_cubemap(c, r)
        assert torch.allclose(s, c, atol=1e-12)


def test_cube_sampling_definition():
    g = torch.Generator().manual_seed(0)
    tex = torch.rand(6, 8, 8, 3, generator=g, dtype=torch.float64)
    D = E.texel_dirs(8, torch.float64)
    # a texel-centre direction returns the texel, for any positive scaling of the direction
    out = E.cube_sample_linear(tex, D.reshape(-1, 3) * 3.7)
    assert torch.allclose(out.reshape(6, 8, 8, 3), tex, atol=1e-12)
    # continuity across a face edge: directions epsilon apart on either side of the +x/+z edge
    a = torch.tensor([[1.0, 0.2, 1.0 - 1e-9], [1.0 - 1e-9, 0.2, 1.0]], dtype=torch.float64)
    o = E.cube_sample_linear(tex, a)
    assert float((o[0] - o[1]).abs().max()) < 0.5  # bounded jump (nearest re-projection, not seamless)
    # mip interpolation: integer levels pick one level, halves average
    mips = [tex, E.cubemap_mip(tex), E.cubemap_mip(E.cubemap_mip(tex))]
    d = torch.nn.functional.normalize(torch.randn(50, 3, generator=g, dtype=torch.float64), dim=-1)
    l1 = E.cube_sample_mip(mips, d, torch.full((50,), 1.0, dtype=torch.float64))
    assert torch.allclose(l1, E.cube_sample_linear(mips[1], d))
    lh = E.cube_sample_mip(mips, d, torch.full((50,), 1.5, dtype=torch.float64))
    assert torch.allclose(lh, 0.5 * (E.cube_sample_linear(mips[1], d) + E.cube_sample_linear(mips[2], d)))
    # clamped above the last level
    lt = E.cube_sample_mip(mips, d, torch.full((50,), 7.0, dtype=torch.float64))
    assert torch.allclose(lt, E.cube_sample_linear(mips[2], d))


def test_stage1_golden_matches_oracle(golden_dir):
    """light.py build_mips / get_mip / eval_mip + texture.py stage 1 (reference code) vs the oracle's
    restatement, same weights: mip chain, 24-channel colours, gradients incl. d/d base."""
    z0, z = load(golden_dir, "texture_stage0.npz"), load(golden_dir, "texture_stage1.npz")
    base = z["base"].double().requires_grad_(True)
    spec, diffuse = E.build_mips(base)
    assert len(spec) == 3
    for i, m in enumerate(spec):
        assert torch.allclose(m.float(), z["spec%d" % i], rtol=1e-5, atol=1e-6)
    assert torch.allclose(diffuse.float(), z["diffuse"], rtol=1e-5, atol=1e-6)
    nets = nets_from({k: v.double() for k, v in z0.items()})
    feats = z0["features"].double().requires_grad_(True)
    nrm = z0["normals"].double().requires_grad_(True)
    col = otex.texture_stage1(
        feats, z0["dirs"].double(), nrm, z0["positions"].double(), nets, otex.synthetic_fg_lut().double(),
        lambda n: E.cube_sample_linear(diffuse, n),
        lambda wo, r: E.cube_sample_mip(spec, wo, E.get_mip(r, len(spec))[:, 0]))
    assert col.shape == (257, 24)
    assert torch.allclose(col.float(), z["colors"], rtol=1e-4, atol=1e-5)
    g_f, g_n, g_b = torch.autograd.grad(col, [feats, nrm, base], z["gcolors"].double())
    assert torch.allclose(g_f.float(), z["g_features"], rtol=1e-3, atol=1e-5)
    assert torch.allclose(g_n.float(), z["g_normals"], rtol=1e-3, atol=1e-4)
    assert torch.allclose(g_b.float(), z["g_base"], rtol=1e-3, atol=1e-7)
