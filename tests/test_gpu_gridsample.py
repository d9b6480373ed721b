"""GPU parity for grid_sample 2-D: forward / backward vs aten (CPU), double backward vs the fp64 explicit
oracle, and the FG-LUT lookup convention."""
import pytest
import torch
import torch.nn.functional as F

from oracle import gridsample as ogs
from helpers import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("pad,align", [("zeros", False), ("border", False), ("zeros", True), ("border", True)])
def test_grid_sample_first_and_second_order(dev, pad, align):
    from rise_sdf_amd.gridsample import grid_sample_2d
    g = torch.Generator().manual_seed(1)
    inp = torch.randn(2, 3, 16, 11, generator=g)
    grid = torch.rand(2, 50, 3, 2, generator=g) * 2.6 - 1.3
    go = torch.randn(2, 3, 50, 3, generator=g)
    # first order vs aten
    ic, gc = inp.clone().requires_grad_(True), grid.clone().requires_grad_(True)
    out_c = F.grid_sample(ic, gc, mode="bilinear", padding_mode=pad, align_corners=align)
    gi_c, gg_c = torch.autograd.grad(out_c, [ic, gc], go)
    ig, gg_ = inp.to(dev).requires_grad_(True), grid.to(dev).requires_grad_(True)
    out_g = grid_sample_2d(ig, gg_, pad, align)
    gi_g, gg_g = torch.autograd.grad(out_g, [ig, gg_], go.to(dev), create_graph=True)
    assert torch.allclose(out_g.cpu(), out_c, rtol=1e-5, atol=1e-5)
    assert torch.allclose(gi_g.detach().cpu(), gi_c, rtol=1e-5, atol=1e-5)
    assert torch.allclose(gg_g.detach().cpu(), gg_c, rtol=1e-4, atol=1e-4)
    # second order: L = <gi, A> + <gg, B>; d L / d (go, input, grid) vs the fp64 explicit oracle
    A, B = torch.randn(gi_c.shape, generator=g), torch.randn(gg_c.shape, generator=g)
    go_g = go.to(dev).requires_grad_(True)
    ig2, gg2 = inp.to(dev).requires_grad_(True), grid.to(dev).requires_grad_(True)
    o2 = grid_sample_2d(ig2, gg2, pad, align)
    gi2, ggr2 = torch.autograd.grad(o2, [ig2, gg2], go_g, create_graph=True)
    L = (gi2 * A.to(dev)).sum() + (ggr2 * B.to(dev)).sum()
    d_go, d_in, d_grid = torch.autograd.grad(L, [go_g, ig2, gg2])
    i64, g64, go64 = inp.double().requires_grad_(True), grid.double().requires_grad_(True), go.double().requires_grad_(True)
    o64 = ogs.grid_sample_2d(i64, g64, pad, align)
    gi64, gg64 = torch.autograd.grad(o64, [i64, g64], go64, create_graph=True)
    L64 = (gi64 * A.double()).sum() + (gg64 * B.double()).sum()
    r_go, r_in, r_grid = torch.autograd.grad(L64, [go64, i64, g64])
    assert rel_err(d_go, r_go) < 1e-5
    assert rel_err(d_in, r_in) < 1e-5
    assert rel_err(d_grid, r_grid) < 1e-4


def test_fg_lut_lookup_convention(dev):
    """Texel centres at (i + 1/2)/size, clamp to edge (models/texture.py:338-341)."""
    from rise_sdf_amd.gridsample import fg_lut_lookup
    lut = torch.arange(8 * 8 * 2, dtype=torch.float32).reshape(1, 8, 8, 2)
    uv = torch.tensor([[(3 + 0.5) / 8, (5 + 0.5) / 8], [0.0, 0.0], [1.0, 1.0], [(3 + 1.0) / 8, (5 + 0.5) / 8]])
    out = fg_lut_lookup(lut.to(dev), uv.to(dev)).cpu()
    assert torch.allclose(out[0], lut[0, 5, 3])          # exact texel centre: u -> x (column), v -> y (row)
    assert torch.allclose(out[1], lut[0, 0, 0]) and torch.allclose(out[2], lut[0, 7, 7])  # clamp
    assert torch.allclose(out[3], (lut[0, 5, 3] + lut[0, 5, 4]) / 2)
