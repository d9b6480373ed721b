"""CPU: the explicit-bilinear oracle agrees with aten's grid_sample to first order (values + gradients),
which validates it as the second-order reference."""
import pytest
import torch
import torch.nn.functional as F

from oracle import gridsample as ogs


@pytest.mark.parametrize("pad,align", [("zeros", False), ("border", False), ("zeros", True), ("border", True)])
def test_explicit_bilinear_matches_aten_first_order(pad, align):
    g = torch.Generator().manual_seed(0)
    inp = torch.randn(2, 3, 7, 9, generator=g, dtype=torch.float64, requires_grad=True)
    grid = (torch.rand(2, 5, 4, 2, generator=g, dtype=torch.float64) * 2.6 - 1.3).requires_grad_(True)
    go = torch.randn(2, 3, 5, 4, generator=g, dtype=torch.float64)
    a = F.grid_sample(inp, grid, mode="bilinear", padding_mode=pad, align_corners=align)
    ga = torch.autograd.grad(a, [inp, grid], go)
    b = ogs.grid_sample_2d(inp, grid, pad, align)
    gb = torch.autograd.grad(b, [inp, grid], go)
    assert torch.allclose(a, b, atol=1e-12)
    assert torch.allclose(ga[0], gb[0], atol=1e-12) and torch.allclose(ga[1], gb[1], atol=1e-12)
