"""``nvdiffrast.torch.texture`` on the HIP lookup kernels (csrc/gridsample.hip, csrc/envlight.hip).

nvdiffrast is neither vendored nor pinned by the reference, so the addressing conventions are this build's
statement of nvdiffrast's documented behaviour: texel centres at
``(i + 1/2) / size``; ``boundary_mode='clamp'`` clamps to the edge texel; ``boundary_mode='cube'`` picks the
major-axis face and re-projects off-face taps; ``filter_mode='linear-mipmap-linear'`` with an explicit ``mip``
stack and no ``uv_da`` uses ``mip_level_bias`` as the (fractional) level.  Only the call shapes the reference
issues are implemented; the rest raise.

  2-D   dr.texture(tex [1,H,W,C], uv [1,S,1,2], filter_mode='linear', boundary_mode='clamp')         texture.py:338-341
  cube  dr.texture(tex [1,6,R,R,C], dirs [1,S,1,3], filter_mode='linear', boundary_mode='cube')       light.py:202-205
  cube  dr.texture(tex [1,6,R,R,C], dirs [1,S,1,3], mip=[[1,6,r,r,C] ...], mip_level_bias=[1,S,1],
                   filter_mode='linear-mipmap-linear', boundary_mode='cube')                             light.py:194-199
"""
from __future__ import annotations

import torch as _torch

from ..envlight import texture_cube as _texture_cube
from ..gridsample import grid_sample_2d as _grid_sample_2d


def texture(tex, uv, uv_da=None, mip_level_bias=None, mip=None, filter_mode="auto", boundary_mode="wrap",
            max_mip_level=None):
    """Same positional order and keyword names as ``nvdiffrast.torch.texture``.  Differentiable w.r.t. ``tex``,
    every tensor of ``mip``, ``uv`` and ``mip_level_bias`` (first order; the 2-D form also second order)."""
    if uv_da is not None or max_mip_level is not None:
        raise NotImplementedError("dr.texture: screen-space derivatives / max_mip_level are not used by RISE-SDF")
    if filter_mode == "auto":
        filter_mode = "linear-mipmap-linear" if mip is not None else "linear"
    if boundary_mode == "cube":
        if tex.ndim != 5 or tex.shape[0] != 1 or tex.shape[1] != 6 or uv.shape[-1] != 3:
            raise ValueError("dr.texture(cube): tex must be [1,6,R,R,C] and uv [1,H,W,3]")
        out_shape = tuple(uv.shape[:-1]) + (tex.shape[-1],)
        dirs = uv.reshape(-1, 3)
        if filter_mode == "linear":
            out = _texture_cube(tex[0], dirs)
        elif filter_mode == "linear-mipmap-linear":
            if mip is None or mip_level_bias is None:
                raise NotImplementedError("dr.texture(cube, mipmapped): RISE-SDF always passes an explicit mip "
                                          "stack and mip_level_bias (lib/pbr/light.py:194-199)")
            out = _texture_cube(tex[0], dirs, mips=[m[0] for m in mip], mip_level_bias=mip_level_bias.reshape(-1))
        else:
            raise NotImplementedError(f"dr.texture(cube): filter_mode={filter_mode!r}")
        return out.reshape(out_shape)
    if boundary_mode == "clamp":
        if filter_mode != "linear" or mip is not None:
            raise NotImplementedError("dr.texture(2-D): only filter_mode='linear' without mips is on the hot path")
        if tex.ndim != 4 or uv.shape[-1] != 2 or tex.shape[0] != uv.shape[0]:
            raise ValueError("dr.texture(2-D): tex must be [N,H,W,C] and uv [N,h,w,2]")
        grid = uv * 2.0 - 1.0
        out = _grid_sample_2d(tex.permute(0, 3, 1, 2).contiguous(), grid, padding_mode="border",
                              align_corners=False)             # [N,C,h,w]
        return out.permute(0, 2, 3, 1)
    raise NotImplementedError(f"dr.texture: boundary_mode={boundary_mode!r} is not used on the RISE-SDF hot path "
                              "(lat-long environment import, light_utils.py:138, is I/O outside it)")
