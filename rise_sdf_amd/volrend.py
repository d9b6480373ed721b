"""Host-side mirror of the two live functions of the reference's ``models/volrend.py``:
``rendering_with_normals_sdf`` (:739-895) and ``secondary_rendering`` (:18-127).  The other five
``rendering_*`` variants of that file are unreferenced by the shipped configs.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import torch
from torch import Tensor

from . import ops


def rendering_with_normals_sdf(t_starts: Tensor, t_ends: Tensor, ray_indices: Optional[Tensor] = None,
                               n_rays: Optional[int] = None, rgb_sigma_fn: Optional[Callable] = None,
                               rgb_alpha_fn: Optional[Callable] = None, render_bkgd: Optional[Tensor] = None,
                               has_laplace: bool = False, color_dim=3, normal_dim=3
                               ) -> Tuple[Tensor, Tensor, Tensor, Tensor, Dict]:
    """-> (colors [N,C], normals [N,3], opacities [N,1], depths [N,1], extras).  Depth is NOT divided by
    opacity (volrend.py:886)."""
    if ray_indices is not None:
        assert t_starts.shape == t_ends.shape == ray_indices.shape, \
            "Since nerfacc 0.5.0, t_starts, t_ends and ray_indices must have the same shape (N,). "
    if rgb_sigma_fn is None and rgb_alpha_fn is None:
        raise ValueError("At least one of `rgb_sigma_fn` and `rgb_alpha_fn` should be specified.")
    if rgb_sigma_fn is not None:
        raise NotImplementedError("rgb_sigma_fn is not implemented yet.")
    dev = t_starts.device
    sdf_laplace = None
    if t_starts.shape[0] != 0:
        res = rgb_alpha_fn(t_starts, t_ends, ray_indices)
        if has_laplace:
            rgbs, normals, alphas, sdf, sdf_grad, sdf_laplace = res
        else:
            rgbs, normals, alphas, sdf, sdf_grad = res
    else:
        rgbs = torch.empty((0, color_dim), device=dev)
        normals = torch.empty((0, normal_dim), device=dev)
        alphas = torch.empty((0,), device=dev)
        sdf = torch.empty((0,), device=dev)
        sdf_grad = torch.empty((0, 3), device=dev)
        sdf_laplace = torch.empty((0, 3), device=dev)
    assert rgbs.shape[-1] == color_dim, f"rgbs must have {color_dim} channels, got {rgbs.shape}"
    assert normals.shape[-1] == normal_dim
    assert alphas.shape == t_starts.shape, f"alphas must have shape of (N,)! Got {alphas.shape}"
    assert sdf.shape == t_starts.shape and sdf_grad.shape[-1] == 3
    packed = ops.pack_info(ray_indices, n_rays)
    weights, trans = ops.render_weight_from_alpha(alphas, packed_info=packed)
    extras = {"weights": weights, "trans": trans, "rgbs": rgbs, "alphas": alphas, "normals": normals,
              "sdf": sdf, "sdf_grad": sdf_grad, "packed_info": packed}
    if has_laplace:
        extras["sdf_laplace"] = sdf_laplace
    colors = ops.accumulate_along_rays(weights, rgbs, packed_info=packed)
    if ops.fold_normals() and normals.shape[-1] == 3:
        # opacity, depth and the normal map in one pass, same bits as the three accumulate calls (volrend.py:875-885)
        opacities, depths, normals_map = ops.accumulate_opacity_depth_normal(weights, t_starts, t_ends, normals,
                                                                             packed_info=packed)
    else:
        normals_map = ops.accumulate_along_rays(weights, normals, packed_info=packed)
        opacities, depths = ops.accumulate_opacity_depth(weights, t_starts, t_ends, packed_info=packed)
    if render_bkgd is not None:
        colors = colors + render_bkgd * (1.0 - opacities)
        normals_map = normals_map + render_bkgd * (1 - opacities) * torch.tensor([0.0, 0.0, 1.0], device=dev)
    return colors, normals_map, opacities, depths, extras


@torch.no_grad()
def secondary_rendering(t_starts: Tensor, t_ends: Tensor, ray_indices: Tensor, n_rays: int,
                        alpha_fn: Callable, chunk_size: int = 160000, phantom_last_ray: bool = False,
                        alphas: Optional[Tensor] = None):
    """Opacity / depth of the secondary (reflection) rays (volrend.py:18-127): alpha in chunks, then
    weights and two accumulations.  -> (opacities [N,1], depths [N,1], extras).
    ``phantom_last_ray``: the last ray only owns the unused tail of capacity-sized sample arrays (the read-free sampling pass of
    models/split_mixed_occ.py's mirror); it gets an empty range, so that the per-ray kernels -- C1 walks a ray's samples in the
    reference's sequential order -- do not composite a hundred thousand dummy samples for a result nobody reads."""
    dev = t_starts.device
    if alphas is not None:
        # the sampler's own alphas of exactly these samples (OccGridEstimator.sampling(..., return_alphas=True)): the reference
        # evaluates the field a second time here (volrend.py:60-75) and gets the same numbers
        assert alphas.shape == t_starts.shape
    elif t_starts.shape[0] != 0:
        alphas = torch.cat([alpha_fn(t_starts[i:i + chunk_size], t_ends[i:i + chunk_size],
                                     ray_indices[i:i + chunk_size])
                            for i in range(0, t_starts.shape[0], chunk_size)], dim=0)
    else:
        alphas = torch.empty((0,), device=dev)
    packed = ops.pack_info(ray_indices, n_rays)
    if phantom_last_ray and n_rays > 0:
        packed = packed.clone()
        packed[n_rays - 1:, 1].zero_()           # (a fill on the device: indexing with a Python scalar would be a host copy)
    weights, trans = ops.render_weight_from_alpha(alphas, packed_info=packed)
    opacities, depths = ops.accumulate_opacity_depth(weights, t_starts, t_ends, packed_info=packed)
    return opacities, depths, {"weights": weights, "trans": trans, "alphas": alphas}
