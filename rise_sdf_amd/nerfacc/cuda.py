"""The vendored nerfacc-0.3.5 ``_C`` extension surface (lib/nerfacc/cuda/csrc/pybind.cu:131-170) on the HIP
kernels: same function names, positional order, shapes and dtypes, so that the reference's own Python in
``lib/nerfacc/*.py`` (ray_marching.py, vol_rendering.py, grid.py, pack.py, intersection.py) runs unchanged
once its lazy loader resolves here:

    sys.modules["lib.nerfacc.cuda._backend"] = rise_sdf_amd.nerfacc.cuda._backend

(``lib/nerfacc/cuda/__init__.py:8-15`` does ``from ._backend import _C`` on first call.)  Entries the hot path
never reaches (``ray_resampling``, ``unpack_data``, non-AABB contraction, the ``*_cub`` variants -- this build
reports ``is_cub_available() == False`` so the packed_info kernels are used) raise NotImplementedError.
"""
from __future__ import annotations

import sys
import types
from enum import Enum

import torch

from .. import ops
from .._lib import check, lib, ptr, require_device, stream_ptr


class ContractionType(Enum):
    """pybind.cu:134-137; ``ContractionType(int)`` is what ``lib/nerfacc/contraction.py:62`` calls."""
    AABB = 0
    UN_BOUNDED_TANH = 1
    UN_BOUNDED_SPHERE = 2


def _aabb_only(ctype):
    if getattr(ctype, "value", ctype) != 0:
        raise NotImplementedError("only ContractionType.AABB is on the RISE-SDF hot path")


def _flat(t):
    return t.detach().to(torch.float32).contiguous().reshape(-1)


# ---- contraction / grid --------------------------------------------------------------------------
def contract(samples, roi, ctype):
    """helpers_contraction.h:16-21 (AABB): (x - min) / (max - min)."""
    _aabb_only(ctype)
    return (samples - roi[:3]) / (roi[3:] - roi[:3])


def contract_inv(samples, roi, ctype):
    """helpers_contraction.h:23-28 (AABB)."""
    _aabb_only(ctype)
    return samples * (roi[3:] - roi[:3]) + roi[:3]


def grid_query(samples, roi, grid_value, ctype):
    """ray_marching.cu:295-358: value of the cell holding each sample, 0 outside the (inclusive) box."""
    _aabb_only(ctype)
    inside, cell = ops.query_occ(samples, roi, torch.ones_like(grid_value, dtype=torch.bool), return_cell=True)
    vals = grid_value.reshape(-1)[cell.long().clamp_(min=0)]
    return torch.where(inside, vals, torch.zeros_like(vals))


# ---- marching --------------------------------------------------------------------------------------
def ray_aabb_intersect(rays_o, rays_d, aabb):
    return list(ops.ray_aabb_intersect(rays_o, rays_d, aabb))


def ray_marching(rays_o, rays_d, t_min, t_max, roi, grid_binary, ctype, step_size, cone_angle):
    """-> [packed_info int32 [N,2], ray_indices int64 [S], t_starts [S,1], t_ends [S,1]] (ray_marching.cu:194-289)."""
    _aabb_only(ctype)
    packed, ri, ts, te = ops.march(rays_o, rays_d, t_min, t_max, roi, grid_binary, step_size, cone_angle)
    return [packed, ri, ts[:, None], te[:, None]]


def ray_resampling(*_a, **_k):
    raise NotImplementedError("cdf.cu ray_resampling is not used by RISE-SDF (SURVEY.md section 2)")


# ---- rendering ---------------------------------------------------------------------------------------
def is_cub_available():
    return False


def _pk(packed_info):
    return packed_info.to(torch.int32).contiguous()


def weight_from_alpha_forward_naive(packed_info, alphas):
    pk, a = _pk(packed_info), _flat(alphas)
    require_device(pk, a)
    w, t = torch.empty_like(a), torch.empty_like(a)
    check(lib().rsdf_weight_from_alpha_fwd(ptr(pk), ptr(a), pk.shape[0], ptr(w), ptr(t), stream_ptr()),
          "weight_from_alpha_fwd")
    return w.view(alphas.shape)


def weight_from_alpha_backward_naive(weights, grad_weights, packed_info, alphas):
    pk, a, w, gw = _pk(packed_info), _flat(alphas), _flat(weights), _flat(grad_weights)
    require_device(pk, a, w, gw)
    # T = w / alpha is not recoverable at alpha = 0: recompute the scan (one extra pass over [S])
    w2, t = torch.empty_like(a), torch.empty_like(a)
    check(lib().rsdf_weight_from_alpha_fwd(ptr(pk), ptr(a), pk.shape[0], ptr(w2), ptr(t), stream_ptr()),
          "weight_from_alpha_fwd")
    ga = torch.empty_like(a)
    check(lib().rsdf_weight_from_alpha_bwd(ptr(pk), ptr(a), ptr(w), ptr(t), ptr(gw), pk.shape[0], ptr(ga),
                                           stream_ptr()), "weight_from_alpha_bwd")
    return ga.view(alphas.shape)


def transmittance_from_alpha_forward_naive(packed_info, alphas):
    pk, a = _pk(packed_info), _flat(alphas)
    require_device(pk, a)
    w, t = torch.empty_like(a), torch.empty_like(a)
    check(lib().rsdf_weight_from_alpha_fwd(ptr(pk), ptr(a), pk.shape[0], ptr(w), ptr(t), stream_ptr()),
          "weight_from_alpha_fwd")
    return t.view(alphas.shape)


def transmittance_from_alpha_backward_naive(packed_info, alphas, transmittance, transmittance_grad):
    pk, a, t, gt = _pk(packed_info), _flat(alphas), _flat(transmittance), _flat(transmittance_grad)
    require_device(pk, a, t, gt)
    ga = torch.empty_like(a)
    check(lib().rsdf_transmittance_from_alpha_bwd(ptr(pk), ptr(a), ptr(t), ptr(gt), pk.shape[0], ptr(ga),
                                                  stream_ptr()), "transmittance_from_alpha_bwd")
    return ga.view(alphas.shape)


def _alpha_of(starts, ends, sigmas):
    return 1.0 - torch.exp(-sigmas * (ends - starts))


def weight_from_sigma_forward_naive(packed_info, starts, ends, sigmas):
    return weight_from_alpha_forward_naive(packed_info, _alpha_of(starts, ends, sigmas))


def weight_from_sigma_backward_naive(weights, grad_weights, packed_info, starts, ends, sigmas):
    a = _alpha_of(starts, ends, sigmas)
    return weight_from_alpha_backward_naive(weights, grad_weights, packed_info, a) * (ends - starts) * (1.0 - a)


def transmittance_from_sigma_forward_naive(packed_info, starts, ends, sigmas):
    return transmittance_from_alpha_forward_naive(packed_info, _alpha_of(starts, ends, sigmas))


def transmittance_from_sigma_backward_naive(packed_info, starts, ends, transmittance, transmittance_grad):
    raise NotImplementedError("density transmittance backward: the learned background is disabled in RISE-SDF's "
                              "configs; use the alpha form")


def _no_cub(*_a, **_k):
    raise NotImplementedError("is_cub_available() is False in this build: the packed_info kernels are used")


transmittance_from_sigma_forward_cub = transmittance_from_sigma_backward_cub = _no_cub
transmittance_from_alpha_forward_cub = transmittance_from_alpha_backward_cub = _no_cub


# ---- pack / unpack -----------------------------------------------------------------------------------
def unpack_info(packed_info, n_samples):
    return ops.unpack_info(packed_info, int(n_samples))


def unpack_info_to_mask(packed_info, n_samples):
    raise NotImplementedError("pack.cu unpack_info_to_mask (only reached from unpack_data) is not on the hot path")


def unpack_data(packed_info, data, n_samples_per_ray):
    raise NotImplementedError("pack.cu unpack_data is not used by RISE-SDF")


# ``from ._backend import _C``: a module named like the reference's loader whose ``_C`` is this module
_backend = types.ModuleType(__name__ + "._backend")
_backend._C = sys.modules[__name__]
