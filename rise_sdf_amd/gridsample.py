"""Bilinear 2-D ``grid_sample`` with first- and second-order gradients on the HIP kernels.

Mirror of the reference's ``utils/cuda_gridsample.py`` (``grid_sample_2d``; same argument meaning and
assertions) and of the plugin entry ``grad2_2d`` (lib/grid_sample_grad2/gridsample_cuda.cpp:26-37).
Unlike the reference -- whose forward/backward call aten and whose double backward is a custom CUDA
kernel -- all three levels run this repo's kernels (csrc/gridsample.hip).

Also the live use: ``fg_lut_lookup`` == ``dr.texture(FG_LUT[1,H,W,2], uv[1,S,1,2], filter_mode='linear',
boundary_mode='clamp')`` of models/texture.py:338-341.
"""
from __future__ import annotations

import torch

from ._lib import check, lib, ptr, require_device, stream_ptr


def _f(t):
    return None if t is None else t.detach().to(torch.float32).contiguous()


def _dims(input, grid):
    N, C, H, W = input.shape
    return N, C, H, W, grid.shape[1], grid.shape[2]


class _GridSample2dForward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, grid, padding_mode, align_corners):
        assert input.ndim == 4
        assert grid.ndim == 4
        assert input.shape[0] == grid.shape[0]
        assert grid.shape[3] == 2
        inp, g = _f(input), _f(grid)
        require_device(inp, g)
        N, C, H, W, Ho, Wo = _dims(inp, g)
        out = torch.empty(N, C, Ho, Wo, dtype=torch.float32, device=inp.device)
        pm = ["zeros", "border"].index(padding_mode)
        check(lib().rsdf_grid_sample2d_fwd(ptr(inp), ptr(g), N, C, H, W, Ho, Wo, pm, int(align_corners), ptr(out),
                                           stream_ptr()), "grid_sample2d_fwd")
        ctx.save_for_backward(input, grid)
        ctx.padding_mode, ctx.align_corners = pm, bool(align_corners)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        input, grid = ctx.saved_tensors
        # a constant texture (the FG LUT is a buffer) needs no gradient: skipping it skips one float atomic per tap into a
        # 256 x 256 map that every sample of the batch hits (34 ms of the c2 step)
        gi, gg = _GridSample2dBackward.apply(grad_output, input, grid, ctx.padding_mode, ctx.align_corners,
                                             bool(ctx.needs_input_grad[0]))
        return gi, gg, None, None


class _GridSample2dBackward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, grad_output, input, grid, padding_mode, align_corners, need_input_grad=True):
        go, inp, g = _f(grad_output), _f(input), _f(grid)
        require_device(go, inp, g)
        N, C, H, W, Ho, Wo = _dims(inp, g)
        grad_input = torch.zeros_like(inp) if need_input_grad else None
        grad_grid = torch.empty_like(g)
        check(lib().rsdf_grid_sample2d_bwd(ptr(go), ptr(inp), ptr(g), N, C, H, W, Ho, Wo, padding_mode,
                                           int(align_corners), ptr(grad_input), ptr(grad_grid), stream_ptr()),
              "grid_sample2d_bwd")
        ctx.save_for_backward(grad_output, input, grid)
        ctx.padding_mode, ctx.align_corners = padding_mode, align_corners
        return grad_input, grad_grid

    @staticmethod
    def backward(ctx, grad2_grad_input, grad2_grad_grid):
        grad_output, input, grid = ctx.saved_tensors
        out = grad2_2d(grad2_grad_input, grad2_grad_grid, grad_output, input, grid, ctx.padding_mode,
                       ctx.align_corners)
        return out[0], out[1], out[2], None, None, None


def grad2_2d(g2_input, g2_grid, grad_output, input, grid, padding_mode, align_corners):
    """lib/grid_sample_grad2/gridsample_cuda.cpp:26-37: -> [grad_grad_output, grad_input, grad_grid].
    padding_mode: 0 zeros, 1 border."""
    g2i, g2g, go, inp, g = _f(g2_input), _f(g2_grid), _f(grad_output), _f(input), _f(grid)
    require_device(g2i, g2g, go, inp, g)
    N, C, H, W, Ho, Wo = _dims(inp, g)
    ggo = torch.empty_like(go)
    gin = torch.zeros_like(inp)
    ggr = torch.empty_like(g)
    check(lib().rsdf_grid_sample2d_bwd2(ptr(g2i), ptr(g2g), ptr(go), ptr(inp), ptr(g), N, C, H, W, Ho, Wo,
                                        int(padding_mode), int(bool(align_corners)), ptr(ggo), ptr(gin), ptr(ggr),
                                        stream_ptr()), "grid_sample2d_bwd2")
    return [ggo, gin, ggr]


def grid_sample_2d(input, grid, padding_mode="zeros", align_corners=True):
    assert padding_mode in ["zeros", "border"]
    return _GridSample2dForward.apply(input, grid, padding_mode, align_corners)


def grid_sample_3d(input, grid, padding_mode="zeros", align_corners=True):
    raise NotImplementedError("no 3-D texture is on the RISE-SDF hot path (SURVEY.md section 2 row 14)")


def fg_lut_lookup(fg_lut, uv):
    """fg_lut [1,H,W,2] (the reference's buffer layout), uv [S,2] in [0,1] (u = NoV -> x, v = roughness -> y)
    -> [S,2]; bilinear, clamp-to-edge, texel centres at (i + 1/2)/size.  Twice differentiable in uv."""
    lut = fg_lut.permute(0, 3, 1, 2).contiguous()
    grid = (uv * 2.0 - 1.0).reshape(1, -1, 1, 2)
    out = grid_sample_2d(lut, grid, padding_mode="border", align_corners=False)  # [1,2,S,1]
    return out[0, :, :, 0].t()
