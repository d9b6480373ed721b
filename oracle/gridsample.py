"""Oracle for S2/S3 (TEST INFRASTRUCTURE ONLY): explicit bilinear sampling in fp64, differentiated once
and twice by autograd.  First order is cross-checked against aten's
``F.grid_sample`` in tests/test_oracle_gridsample.py; torch's CPU grid_sample has no double backward,
so the explicit form is the second-order reference.  Coordinate conventions follow
lib/grid_sample_grad2/gridsample_cuda.cu:87-127 (aten's): unnormalise, border => clip with zero
gradient outside, zeros => out-of-range corners contribute nothing."""
from __future__ import annotations

import torch


def grid_sample_2d(input, grid, padding_mode="zeros", align_corners=False):
    """input [N,C,H,W], grid [N,Ho,Wo,2] -> [N,C,Ho,Wo] (any float dtype; use float64 for grad2)."""
    N, C, H, W = input.shape
    gx, gy = grid[..., 0], grid[..., 1]
    if align_corners:
        ix, iy = (gx + 1) / 2 * (W - 1), (gy + 1) / 2 * (H - 1)
    else:
        ix, iy = ((gx + 1) * W - 1) / 2, ((gy + 1) * H - 1) / 2
    if padding_mode == "border":
        ix, iy = ix.clamp(0, W - 1), iy.clamp(0, H - 1)   # clamp: zero gradient where clipped
    x0, y0 = torch.floor(ix.detach()), torch.floor(iy.detach())
    tx, ty = ix - x0, iy - y0
    out = 0
    flat = input.reshape(N, C, H * W)
    for dy, wy in ((0, 1 - ty), (1, ty)):
        for dx, wx in ((0, 1 - tx), (1, tx)):
            xs, ys = (x0 + dx).long(), (y0 + dy).long()
            ok = (xs >= 0) & (xs < W) & (ys >= 0) & (ys < H)
            idx = (ys.clamp(0, H - 1) * W + xs.clamp(0, W - 1)).reshape(N, 1, -1).expand(N, C, -1)
            v = torch.gather(flat, 2, idx).reshape(N, C, *gx.shape[1:])
            out = out + v * (wx * wy * ok)[:, None]
    return out
