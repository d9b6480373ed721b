"""Drop-in for ``from lib import renderutils as ru`` (lib/pbr/light.py:10): the two entry points the hot path
calls (``ru.diffuse_cubemap``, ``ru.specular_cubemap``; lib/renderutils/ops.py:391-458) and a plugin-shaped
object with the names ``lib/renderutils/c_src/torch_bindings.cpp:1053-1057`` exports, so the reference's own
``ops.py`` can also run on it by replacing ``_get_plugin``."""
from __future__ import annotations

import torch

from . import envlight as _E
from ._lib import check, lib, ptr, require_device, stream_ptr


def diffuse_cubemap(cubemap, use_python=False):
    """lib/renderutils/ops.py:404-411 (``use_python`` selects the reference's slow torch path; ignored)."""
    return _E.diffuse_cubemap(cubemap)


def specular_cubemap(cubemap, roughness, cutoff=0.99, use_python=False):
    """lib/renderutils/ops.py:446-458 -> [6,R,R,3] (already divided by the accumulated weight)."""
    return _E.specular_cubemap(cubemap, roughness, cutoff)


class _Plugin:
    """``renderutils_plugin`` surface for the cube-map prefilters (NHWC [6,R,R,3] fp32 contiguous device tensors,
    torch_bindings.cpp:27-31,740-890).  ``bounds`` is this build's own [6,R,R,24] table: opaque to callers, who
    only pass it back."""

    @staticmethod
    def diffuse_cubemap_fwd(cubemap):
        c = cubemap.detach().float().contiguous()
        require_device(c)
        out = torch.empty_like(c)
        check(lib().rsdf_diffuse_cubemap_fwd(ptr(c), c.shape[1], ptr(out), stream_ptr()), "diffuse_cubemap_fwd")
        return out

    @staticmethod
    def diffuse_cubemap_bwd(cubemap, grad):
        g = grad.detach().float().contiguous()
        require_device(g)
        gc = torch.empty_like(g)
        check(lib().rsdf_diffuse_cubemap_bwd(ptr(g), g.shape[1], ptr(gc), stream_ptr()), "diffuse_cubemap_bwd")
        return gc

    @staticmethod
    def specular_bounds(res, costheta_cutoff):
        dev = torch.device("cuda", torch.cuda.current_device())
        b = torch.empty(6, res, res, 24, dtype=torch.float32, device=dev)
        check(lib().rsdf_specular_bounds(int(res), float(costheta_cutoff), ptr(b), stream_ptr()), "specular_bounds")
        return b

    @staticmethod
    def specular_cubemap_fwd(cubemap, bounds, roughness, costheta_cutoff):
        c = cubemap.detach().float().contiguous()
        require_device(c, bounds)
        R = c.shape[1]
        table = _E.texel_table(R, c.device)
        out = torch.empty(6, R, R, 4, dtype=torch.float32, device=c.device)
        check(lib().rsdf_specular_cubemap_fwd(ptr(c), ptr(bounds), ptr(table), R, float(roughness),
                                              float(costheta_cutoff), ptr(out), stream_ptr()), "specular_cubemap_fwd")
        return out

    @staticmethod
    def specular_cubemap_bwd(cubemap, bounds, grad, roughness, costheta_cutoff):
        g = grad.detach().float().contiguous()
        require_device(g, bounds)
        R = g.shape[1]
        table = _E.texel_table(R, g.device)
        gc = torch.empty(6, R, R, 3, dtype=torch.float32, device=g.device)
        check(lib().rsdf_specular_cubemap_bwd(ptr(g), g.shape[-1], ptr(bounds), ptr(table), R, float(roughness),
                                              float(costheta_cutoff), ptr(gc), stream_ptr()), "specular_cubemap_bwd")
        return gc


plugin = _Plugin()


def _get_plugin():
    return plugin


__all__ = ["diffuse_cubemap", "specular_cubemap", "plugin"]
