"""``nerfacc.volrend`` of nerfacc 0.5.3: the names models/volrend.py:10-14 imports."""
from . import accumulate_along_rays, render_weight_from_alpha, render_weight_from_density  # noqa: F401
from . import render_transmittance_from_alpha, render_visibility  # noqa: F401

__all__ = ["render_weight_from_density", "render_weight_from_alpha", "accumulate_along_rays",
           "render_transmittance_from_alpha", "render_visibility"]
