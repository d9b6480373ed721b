"""CPU oracle for the RISE-SDF ray-marched SDF hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this package.  The shipped path (``rise_sdf_amd``) never does and raises
when its HIP library is missing.

Two halves:

* ``risesdf_oracle.c`` -- plain C for the per-ray / per-sample loops whose integer
  results must be bit exact (marcher, occupancy index, hash-grid indices) and for the
  serial compositing recurrences.  Built by ``oracle/Makefile`` into
  ``oracle/_build/liboracle.so`` and bound here with ctypes.
* fp32 PyTorch-on-CPU restatements of the floating-point layers (weight-normalised
  MLP, finite-difference SDF gradient, NeuS alpha, accumulate).  These use the same
  torch ops the reference's Python uses, so they are pinned against golden vectors
  generated from the imported reference (``tests/golden/make_golden.py``).

Every function cites the reference file:line it follows (paths relative to the
upstream RISE-SDF tree).

Parity pin status (see DESIGN.md "Oracle"):
  C1 compositing ........ pinned: docstring KATs lib/nerfacc/vol_rendering.py:303-307,430-434,493-500
  H3/H4/A1/P1/I0 ........ pinned: golden vectors from the imported reference Python
  M1/M3/M4 marcher ...... restatement + hand-derived cases only (vendored CUDA unbuildable here)
  H1 hash grid .......... PARITY UNPINNED (tiny-cuda-nn absent and unpinned upstream)
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess
from typing import Callable, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None


def build() -> str:
    """Compile the C half (gcc only; seconds)."""
    src = os.path.join(_HERE, "risesdf_oracle.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class GridMeta(ctypes.Structure):
    _fields_ = [
        ("n_levels", ctypes.c_uint32),
        ("n_features", ctypes.c_uint32),
        ("scale", ctypes.c_float * 32),
        ("res", ctypes.c_uint32 * 32),
        ("offset", ctypes.c_uint32 * 32),
        ("size", ctypes.c_uint32 * 32),
    ]


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(t: torch.Tensor):
    return ctypes.c_void_p(t.data_ptr())


def _f32(t):
    return t.detach().to(torch.float32).contiguous()


# --------------------------------------------------------------------------------------
# I0: ray batch.  models/ray_utils.py:9-56; systems/split_occ.py:103
# --------------------------------------------------------------------------------------
def get_ray_directions(W, H, fx, fy, cx, cy):
    """Pinhole directions, pixel centres at +0.5, OpenGL (-z forward, +y up)."""
    u = torch.arange(W, dtype=torch.float32) + 0.5
    v = torch.arange(H, dtype=torch.float32) + 0.5
    vv, uu = torch.meshgrid(v, u, indexing="ij")  # (H, W)
    return torch.stack([(uu - cx) / fx, -(vv - cy) / fy, -torch.ones_like(uu)], -1)


def get_rays(directions: torch.Tensor, c2w: torch.Tensor):
    """models/ray_utils.py:32-56.  directions (H,W,3) or (N,3); c2w (3,4) or, with (N,3) directions, a per-ray
    stack (N,3,4) / (1,3,4).  Returns flat (N,3) origins, directions."""
    d = directions.reshape(-1, 3)
    m = c2w if c2w.ndim == 3 else c2w[None]
    rays_d = (d[:, None, :] * m[:, :3, :3]).sum(-1)
    rays_o = m[:, :, 3].expand(rays_d.shape)
    return rays_o.contiguous(), rays_d.contiguous()


def make_rays(rays_o, rays_d):
    """rays = cat(o, normalize(d))  (systems/split_occ.py:103)."""
    return torch.cat([rays_o, F.normalize(rays_d, p=2, dim=-1)], dim=-1)


# --------------------------------------------------------------------------------------
# M1: lib/nerfacc/cuda/csrc/intersection.cu:16-91
# --------------------------------------------------------------------------------------
def ray_aabb_intersect(rays_o, rays_d, aabb):
    o, d, a = _f32(rays_o), _f32(rays_d), _f32(aabb)
    n = o.shape[0]
    t_min = torch.empty(n, dtype=torch.float32)
    t_max = torch.empty(n, dtype=torch.float32)
    lib().orc_ray_aabb_intersect(ctypes.c_int64(n), _p(o), _p(d), _p(a), _p(t_min), _p(t_max))
    return t_min, t_max


# --------------------------------------------------------------------------------------
# M3: ray_marching.cu:16-45, 295-358
# --------------------------------------------------------------------------------------
def query_occ(samples, roi, binary):
    """Returns (occupied bool [S], cell index int32 [S], -1 outside the box)."""
    x, r = _f32(samples), _f32(roi)
    b = binary.to(torch.uint8).contiguous()
    res = (ctypes.c_int * 3)(*b.shape)
    n = x.shape[0]
    occ = torch.empty(n, dtype=torch.uint8)
    cell = torch.empty(n, dtype=torch.int32)
    lib().orc_query_occ(ctypes.c_int64(n), _p(x), _p(r), res, _p(b), _p(occ), _p(cell))
    return occ.bool(), cell


# --------------------------------------------------------------------------------------
# M4: ray_marching.cu:81-289 (two-pass) + policy lib/nerfacc/ray_marching.py:145-190
# --------------------------------------------------------------------------------------
def ray_marching_packed(rays_o, rays_d, t_min, t_max, roi, binary, step_size, cone_angle=0.0):
    """The native two-pass marcher.  Returns packed_info int32 [N,2], ray_indices int64 [S],
    t_starts, t_ends fp32 [S]."""
    o, d, tn, tf, r = _f32(rays_o), _f32(rays_d), _f32(t_min), _f32(t_max), _f32(roi)
    b = binary.to(torch.uint8).contiguous()
    res = (ctypes.c_int * 3)(*b.shape)
    n = o.shape[0]
    num = torch.zeros(n, dtype=torch.int32)
    args = (ctypes.c_int64(n), _p(o), _p(d), _p(tn), _p(tf), _p(r), res, _p(b),
            ctypes.c_float(step_size), ctypes.c_float(cone_angle))
    lib().orc_ray_marching(*args, None, _p(num), None, None, None)
    cum = torch.cumsum(num, 0, dtype=torch.int32)
    packed = torch.stack([cum - num, num], 1).contiguous()
    total = int(cum[-1]) if n else 0
    ri = torch.empty(total, dtype=torch.int64)
    ts = torch.empty(total, dtype=torch.float32)
    te = torch.empty(total, dtype=torch.float32)
    lib().orc_ray_marching(*args, _p(packed), None, _p(ri), _p(ts), _p(te))
    return packed, ri, ts, te


def ray_marching(rays_o, rays_d, *, scene_aabb=None, grid_roi=None, grid_binary=None,
                 t_min=None, t_max=None, near_plane=None, far_plane=None,
                 render_step_size=1e-3, stratified_u=None, cone_angle=0.0,
                 alpha_fn: Optional[Callable] = None, early_stop_eps=1e-4, alpha_thre=0.0):
    """Python-level policy of lib/nerfacc/ray_marching.py:145-220.

    ``stratified_u`` is the explicit U[0,1) jitter tensor [N] (the reference draws it on
    device, :157-158); None means no jitter.  Dense fallback (no grid) uses a 1x1x1
    all-true grid over +-1e10 (:165-174)."""
    if t_min is None or t_max is None:
        if scene_aabb is not None:
            t_min, t_max = ray_aabb_intersect(rays_o, rays_d, scene_aabb)
        else:
            t_min = torch.zeros(rays_o.shape[0])
            t_max = torch.full((rays_o.shape[0],), 1e10)
    if near_plane is not None:
        t_min = torch.clamp(t_min, min=near_plane)
    if far_plane is not None:
        t_max = torch.clamp(t_max, max=far_plane)
    if stratified_u is not None:
        t_min = t_min + _f32(stratified_u) * render_step_size
    if grid_binary is None:
        grid_roi = torch.tensor([-1e10] * 3 + [1e10] * 3, dtype=torch.float32)
        grid_binary = torch.ones(1, 1, 1, dtype=torch.bool)
    packed, ri, ts, te = ray_marching_packed(rays_o, rays_d, t_min, t_max, grid_roi,
                                             grid_binary, render_step_size, cone_angle)
    if alpha_fn is not None:
        alphas = alpha_fn(ts, te, ri)
        keep = render_visibility(alphas, packed_info=packed, early_stop_eps=early_stop_eps,
                                 alpha_thre=alpha_thre)
        ri, ts, te = ri[keep], ts[keep], te[keep]
    return ri, ts, te


# --------------------------------------------------------------------------------------
# M6: lib/nerfacc/pack.py:47-78, pack.cu:7-28
# --------------------------------------------------------------------------------------
def pack_info(ray_indices, n_rays):
    ri = ray_indices.to(torch.int64).contiguous()
    packed = torch.empty(n_rays, 2, dtype=torch.int32)
    lib().orc_pack_info(ctypes.c_int64(ri.numel()), _p(ri), ctypes.c_int64(n_rays), _p(packed))
    return packed


def unpack_info(packed_info, n_samples):
    pk = packed_info.to(torch.int32).contiguous()
    ri = torch.empty(n_samples, dtype=torch.int64)
    lib().orc_unpack_info(ctypes.c_int64(pk.shape[0]), _p(pk), _p(ri))
    return ri


# --------------------------------------------------------------------------------------
# C1: render_transmittance.cu:85-145, render_weight.cu:86-153 as autograd Functions
# --------------------------------------------------------------------------------------
# The reference binary is nvcc -O3 with the default --fmad=true (lib/nerfacc/cuda/_backend.py:43-44): the weight backward's
# multiply-add pairs are fused (oracle/risesdf_oracle.c).  False = every source operation rounded once (tests set it to
# compare both sequences with the HIP kernel's two modes).
C1_FMAD = True


class _WeightFromAlpha(torch.autograd.Function):
    @staticmethod
    def forward(ctx, packed_info, alphas):
        a = _f32(alphas)
        w = torch.empty_like(a)
        lib().orc_weight_from_alpha_fwd(ctypes.c_int64(packed_info.shape[0]), _p(packed_info),
                                        _p(a), _p(w))
        ctx.save_for_backward(packed_info, a, w)
        return w

    @staticmethod
    def backward(ctx, gw):
        packed_info, a, w = ctx.saved_tensors
        gw = _f32(gw)
        ga = torch.empty_like(a)
        lib().orc_weight_from_alpha_bwd(ctypes.c_int64(packed_info.shape[0]), _p(packed_info),
                                        _p(a), _p(w), _p(gw), ctypes.c_int(1 if C1_FMAD else 0), _p(ga))
        return None, ga


class _TransFromAlpha(torch.autograd.Function):
    @staticmethod
    def forward(ctx, packed_info, alphas):
        a = _f32(alphas)
        t = torch.empty_like(a)
        lib().orc_transmittance_from_alpha_fwd(ctypes.c_int64(packed_info.shape[0]),
                                               _p(packed_info), _p(a), _p(t))
        ctx.save_for_backward(packed_info, a, t)
        return t

    @staticmethod
    def backward(ctx, gt):
        packed_info, a, t = ctx.saved_tensors
        gt = _f32(gt)
        ga = torch.empty_like(a)
        lib().orc_transmittance_from_alpha_bwd(ctypes.c_int64(packed_info.shape[0]),
                                               _p(packed_info), _p(a), _p(t), _p(gt), _p(ga))
        return None, ga


def render_transmittance_from_alpha(alphas, *, ray_indices=None, packed_info=None, n_rays=None):
    if packed_info is None:
        packed_info = pack_info(ray_indices, n_rays)
    return _TransFromAlpha.apply(packed_info.to(torch.int32).contiguous(), alphas)


def render_weight_from_alpha(alphas, *, ray_indices=None, packed_info=None, n_rays=None):
    """nerfacc 0.5.3 call shape (models/volrend.py:851-855): returns (weights, trans)."""
    if packed_info is None:
        packed_info = pack_info(ray_indices, n_rays)
    pk = packed_info.to(torch.int32).contiguous()
    w = _WeightFromAlpha.apply(pk, alphas)
    with torch.no_grad():
        t = _TransFromAlpha.apply(pk, alphas)
    return w, t


def render_visibility(alphas, *, ray_indices=None, packed_info=None, n_rays=None,
                      early_stop_eps=1e-4, alpha_thre=0.0):
    """lib/nerfacc/vol_rendering.py:503-520."""
    with torch.no_grad():
        t = render_transmittance_from_alpha(alphas, ray_indices=ray_indices,
                                            packed_info=packed_info, n_rays=n_rays)
        vis = t >= early_stop_eps
        if alpha_thre > 0:
            vis = vis & (alphas >= alpha_thre)
    return vis


# --------------------------------------------------------------------------------------
# C2: lib/nerfacc/vol_rendering.py:174-198
# --------------------------------------------------------------------------------------
def accumulate_along_rays(weights, values=None, *, ray_indices, n_rays):
    """weights [S]; values [S,D] or None -> [n_rays, D or 1]."""
    src = weights[:, None] if values is None else weights[:, None] * values
    out = torch.zeros(n_rays, src.shape[-1], dtype=src.dtype)
    if ray_indices.numel() == 0:
        return out
    return out.index_add(0, ray_indices.to(torch.int64), src)


# --------------------------------------------------------------------------------------
# H1: hash grid.  Call sites models/network_utils.py:47-50,59.  PARITY UNPINNED (tcnn absent).
# --------------------------------------------------------------------------------------
def grid_meta(n_levels=16, n_features=2, log2_hashmap_size=19, base_resolution=32,
              per_level_scale=1.447269237440378) -> Tuple[GridMeta, int]:
    """Level table.  scale_l is computed in fp64 and rounded once to fp32;
    res_l = ceil(scale_l) + 1; size_l = min(round_up(res_l^3, 8), 2^log2_hashmap_size)."""
    m = GridMeta()
    m.n_levels, m.n_features = n_levels, n_features
    off = 0
    for l in range(n_levels):
        scale = float(np.float32(2.0 ** (l * math.log2(per_level_scale)) * base_resolution - 1.0))
        res = int(math.ceil(scale)) + 1
        size = min(((res ** 3 + 7) // 8) * 8, 1 << log2_hashmap_size)
        m.scale[l], m.res[l], m.offset[l], m.size[l] = scale, res, off, size
        off += size
    return m, off * n_features


class _HashGrid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, table, meta):
        x32, t32 = _f32(x), _f32(table)
        n = x32.shape[0]
        out = torch.empty(n, meta.n_levels * meta.n_features, dtype=torch.float32)
        lib().orc_hashgrid_fwd(ctypes.c_int64(n), _p(x32), _p(t32), ctypes.byref(meta), _p(out))
        ctx.save_for_backward(x32)
        ctx.meta, ctx.n_params = meta, t32.numel()
        return out

    @staticmethod
    def backward(ctx, gout):
        (x32,) = ctx.saved_tensors
        g = _f32(gout)
        dt = torch.zeros(ctx.n_params, dtype=torch.float64)
        lib().orc_hashgrid_bwd(ctypes.c_int64(x32.shape[0]), _p(x32), _p(g),
                               ctypes.byref(ctx.meta), _p(dt))
        return None, dt.to(torch.float32), None


def hashgrid_encode(x, table, meta):
    """x [S,3] in [0,1] -> [S, L*F]; differentiable w.r.t. ``table`` only (FD-normal configs
    never need d/dx; analytic-normal mode is a later row)."""
    return _HashGrid.apply(x, table, meta)


def hashgrid_indices(x, meta):
    x32 = _f32(x)
    idx = torch.empty(x32.shape[0], meta.n_levels, 8, dtype=torch.int32)
    lib().orc_hashgrid_indices(ctypes.c_int64(x32.shape[0]), _p(x32), ctypes.byref(meta), _p(idx))
    return idx


# --------------------------------------------------------------------------------------
# H2: models/network_utils.py:58-68 (progressive mask) and :78-79 (include_xyz)
# --------------------------------------------------------------------------------------
def progressive_level(global_step, start_level, start_step, update_steps, n_levels):
    return min(start_level + max(global_step - start_step, 0) // update_steps, n_levels)


def composite_encoding(x_unit, table, meta, n_active_levels=None, include_xyz=True,
                       xyz_scale=2.0, xyz_offset=-1.0):
    enc = hashgrid_encode(x_unit, table, meta)
    if n_active_levels is not None:
        mask = torch.zeros(meta.n_levels * meta.n_features)
        mask[: n_active_levels * meta.n_features] = 1.0
        enc = enc * mask
    if include_xyz:
        enc = torch.cat([x_unit * xyz_scale + xyz_offset, enc], dim=-1)
    return enc


# --------------------------------------------------------------------------------------
# H3: VanillaMLP.  models/network_utils.py:109-157
# --------------------------------------------------------------------------------------
def sphere_init_mlp_params(dim_in, dim_out, n_neurons, n_hidden_layers, radius=0.5, seed=0):
    """Weight-normalised parameters {g_i [out,1], v_i [out,in], b_i [out]} with the reference's
    sphere initialisation (:130-144).  nn.utils.weight_norm sets g = ||v|| row-wise at wrap time."""
    gen = torch.Generator().manual_seed(seed)
    dims = [dim_in] + [n_neurons] * n_hidden_layers + [dim_out]
    params = []
    for i in range(len(dims) - 1):
        fi, fo = dims[i], dims[i + 1]
        is_first, is_last = i == 0, i == len(dims) - 2
        if is_last:
            b = torch.full((fo,), -radius)
            v = torch.randn(fo, fi, generator=gen) * 1e-4 + math.sqrt(math.pi) / math.sqrt(fi)
        elif is_first:
            b = torch.zeros(fo)
            v = torch.zeros(fo, fi)
            v[:, :3] = torch.randn(fo, 3, generator=gen) * (math.sqrt(2) / math.sqrt(fo))
        else:
            b = torch.zeros(fo)
            v = torch.randn(fo, fi, generator=gen) * (math.sqrt(2) / math.sqrt(fo))
        g = v.norm(dim=1, keepdim=True)
        params.append({"g": g, "v": v, "b": b})
    return params


def weight_norm_effective(g, v):
    """W = g * v / ||v||_row  (torch.nn.utils.weight_norm, dim=0)."""
    return v * (g / v.norm(dim=1, keepdim=True))


# ---- config[4]'s bf16 MLP mode (BASELINE.json configs[4]): the same networks with every matrix operand rounded once to
# bf16 (round to nearest even) and fp32 accumulation, forward and backward -- what one v_mfma_f32_*_bf16 product per
# k-step computes.  ``with oracle.mlp_precision("bf16"):`` switches every nn.Linear restated here (vanilla_mlp,
# texture.relu_mlp) to it; parameters, biases, activations and accumulators stay fp32.
_MLP_PRECISION = ["fp32"]


class mlp_precision:
    def __init__(self, precision):
        assert precision in ("fp32", "bf16", "fp16")
        self.p = precision

    def __enter__(self):
        self.old = _MLP_PRECISION[0]
        _MLP_PRECISION[0] = self.p

    def __exit__(self, *a):
        _MLP_PRECISION[0] = self.old


def _rb(t):
    if _MLP_PRECISION[0] == "fp16":
        # the one-part form of csrc/mlp_x2.hip: operands rounded once to fp16 (the kernels' power-of-two class scales do not
        # change that rounding inside the fp16 normal range; below it the kernels keep MORE bits than this emulation)
        return t.to(torch.float16).to(torch.float32)
    return t.to(torch.bfloat16).to(torch.float32)


class _LinearBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        xr, wr = _rb(x), _rb(w)
        ctx.save_for_backward(xr, wr)
        return xr @ wr.t() + b

    @staticmethod
    def backward(ctx, dy):
        xr, wr = ctx.saved_tensors
        dr = _rb(dy)                                  # dz = dy * act'(y) is formed in fp32, then rounded as an operand
        return dr @ wr, dr.t() @ xr, dy.sum(0)


def linear(x, w, b):
    """nn.Linear at the oracle's current MLP precision."""
    if _MLP_PRECISION[0] in ("bf16", "fp16"):
        return _LinearBF16.apply(x.float(), w.float(), b.float())
    return F.linear(x, w, b)


def vanilla_mlp(x, params, activation="softplus100"):
    """Linear -> act -> ... -> Linear.  ``params`` is a list of {g,v,b} (weight-normed) or
    {w,b}.  Softplus(beta=100, threshold=20) for sphere-init nets, ReLU otherwise (:152-157)."""
    h = x.float()
    for i, p in enumerate(params):
        w = weight_norm_effective(p["g"], p["v"]) if "g" in p else p["w"]
        h = linear(h, w, p["b"])
        if i < len(params) - 1:
            h = F.softplus(h, beta=100) if activation == "softplus100" else F.relu(h)
    return h


# --------------------------------------------------------------------------------------
# P1 + H4: VolumeSDF.forward, finite-difference gradient.  models/geometry.py:206-244,
#          contract_to_unisphere :17-19, scale_anything models/utils.py:109-114
# --------------------------------------------------------------------------------------
def contract_aabb(x, radius):
    return (x - (-radius)) / (radius - (-radius)) * (1 - 0) + 0


def volume_sdf(points, table, meta, mlp_params, *, radius, fd_eps, n_active_levels=None,
               with_grad=True, sdf7_given=None, return_sdf7=False):
    """Returns (sdf [S], grad [S,3] or None, feature [S,D]).  FD taps are clamped to +-radius
    before contraction (geometry.py:241).

    ``sdf7_given`` [S,7] (centre, +x,-x,+y,-y,+z,-z): the VALUES of the seven SDF evaluations are replaced by these
    while the autograd graph stays this function's own (value substitution: x + (given - x).detach()).  The
    finite-difference divide amplifies one ulp of SDF disagreement between two fp32 implementations by 1/eps; handing
    the implementation under test's stencil values in removes that forward amplification, so everything downstream
    of the divide -- normals, alpha, weights, and every gradient back to the table and the weights -- can be compared
    at the tolerances of SURVEY 8(d) (1e-4 on MLP parameters, 1e-3 on table rows) instead of 1/eps times looser."""
    def field(p_unit):
        enc = composite_encoding(p_unit.reshape(-1, 3), table, meta, n_active_levels)
        return vanilla_mlp(enc, mlp_params)

    out = field(contract_aabb(points, radius))
    sdf, feature = out[..., 0], out
    own7 = [sdf]
    if sdf7_given is not None:
        sdf = sdf + (sdf7_given[:, 0].to(sdf.dtype) - sdf).detach()
        feature = torch.cat([sdf[:, None], out[..., 1:]], -1)
    grad = None
    if with_grad:
        eps = fd_eps
        offs = torch.tensor([[eps, 0, 0], [-eps, 0, 0], [0, eps, 0], [0, -eps, 0],
                             [0, 0, eps], [0, 0, -eps]], dtype=points.dtype)
        pd = (points[:, None, :] + offs).clamp(-radius, radius)
        sd = field(contract_aabb(pd, radius))[..., 0].view(-1, 6)
        own7.append(sd)
        if sdf7_given is not None:
            sd = sd + (sdf7_given[:, 1:].to(sd.dtype) - sd).detach()
        grad = 0.5 * (sd[:, 0::2] - sd[:, 1::2]) / eps
    if return_sdf7:      # this function's OWN seven values (before any substitution), [S,7]
        return sdf, grad, feature, torch.cat([own7[0][:, None]] + own7[1:], -1).detach()
    return sdf, grad, feature


def progressive_fd_eps(radius, base_resolution, per_level_scale, current_level):
    """geometry.py:304-318."""
    return 2 * radius / (base_resolution * per_level_scale ** (current_level - 1))


# --------------------------------------------------------------------------------------
# A1: NeuS alpha.  models/split_mixed_occ.py:151-177 (= models/neus.py:128-150)
# --------------------------------------------------------------------------------------
def inv_s_from_variance(variance):
    return torch.exp(variance * 10.0).clip(1e-6, 1e6)


def get_alpha(sdf, normal, dirs, dists, inv_s, cos_anneal_ratio=1.0):
    true_cos = (dirs * normal).sum(-1, keepdim=True)
    iter_cos = -(F.relu(-true_cos * 0.5 + 0.5) * (1.0 - cos_anneal_ratio)
                 + F.relu(-true_cos) * cos_anneal_ratio)
    half = iter_cos * dists.reshape(-1, 1) * 0.5
    prev_cdf = torch.sigmoid((sdf[:, None] - half) * inv_s)
    next_cdf = torch.sigmoid((sdf[:, None] + half) * inv_s)
    p, c = prev_cdf - next_cdf, prev_cdf
    return ((p + 1e-5) / (c + 1e-5)).view(-1).clip(0.0, 1.0)


def occ_alpha(sdf, inv_s, render_step_size):
    """A2: occ_eval_fn, split_mixed_occ.py:108-119 (cos == -1, delta == step)."""
    prev_cdf = torch.sigmoid((sdf[:, None] + render_step_size * 0.5) * inv_s)
    next_cdf = torch.sigmoid((sdf[:, None] - render_step_size * 0.5) * inv_s)
    return ((prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)).view(-1, 1).clip(0.0, 1.0)


# --------------------------------------------------------------------------------------
# M2: occupancy EMA update.  lib/nerfacc/grid.py:196-239
# --------------------------------------------------------------------------------------
def occ_grid_update(occs, indices, occ, resolution, occ_thre=0.01, ema_decay=0.95):
    occs = occs.clone()
    occs[indices] = torch.maximum(occs[indices] * ema_decay, occ)
    binary = (occs > torch.clamp(occs.mean(), max=occ_thre)).view(*resolution)
    return occs, binary


# --------------------------------------------------------------------------------------
# C3 + config[1]: NeuS geometry render.  models/neus.py:227-317 with the FD-normal alpha_fn of
# models/split_mixed_occ.py:228-262 and compositing of models/volrend.py:851-886
# --------------------------------------------------------------------------------------
def neus_geometry_render(rays, ray_indices, t_starts, t_ends, table, meta, mlp_params,
                         variance, *, radius, fd_eps, cos_anneal_ratio=1.0,
                         n_active_levels=None, sdf7_given=None, alphas_given=None):
    """Field query + NeuS alpha + composite for an already-marched sample set.
    Returns dict(opacity [N,1], depth [N,1], comp_normal [N,3], weights, alphas, sdf, sdf_grad, sdf7 (own values),
    alphas_own).  ``sdf7_given``: see volume_sdf.  ``alphas_given`` [S]: the VALUES of alpha are replaced by these (the
    gradient still flows through this function's own get_alpha): the weight backward of render_weight.cu:139-151 divides a
    rounding residue by max(1 - alpha, 1e-10), so an implementation under test can only be compared downstream of it on
    bit-identical alphas (tests/test_gpu_late_regime.py)."""
    n_rays = rays.shape[0]
    rays_o, rays_d = rays[:, 0:3], rays[:, 3:6]
    ri = ray_indices.to(torch.int64)
    t_o, t_d = rays_o[ri], rays_d[ri]
    mid = (t_starts + t_ends)[:, None] / 2.0
    positions = t_o + t_d * mid
    dists = (t_ends - t_starts)[:, None]
    sdf, grad, feature, own7 = volume_sdf(positions, table, meta, mlp_params, radius=radius,
                                          fd_eps=fd_eps, n_active_levels=n_active_levels,
                                          sdf7_given=sdf7_given, return_sdf7=True)
    normal = F.normalize(grad, p=2, dim=-1, eps=1e-6)
    alphas = get_alpha(sdf, normal, t_d, dists, inv_s_from_variance(variance), cos_anneal_ratio)
    alphas_own = alphas.detach().clone()
    if alphas_given is not None:
        alphas = alphas + (alphas_given.to(alphas.dtype) - alphas).detach()
    weights, trans = render_weight_from_alpha(alphas, ray_indices=ri, n_rays=n_rays)
    opacity = accumulate_along_rays(weights, None, ray_indices=ri, n_rays=n_rays)
    depth = accumulate_along_rays(weights, mid, ray_indices=ri, n_rays=n_rays)
    comp_normal = accumulate_along_rays(weights, normal, ray_indices=ri, n_rays=n_rays)
    return {"opacity": opacity, "depth": depth, "comp_normal": comp_normal, "weights": weights,
            "trans": trans, "alphas": alphas, "sdf": sdf, "sdf_grad": grad, "feature": feature,
            "normal": normal, "sdf7": own7, "alphas_own": alphas_own}


# --------------------------------------------------------------------------------------
# N1: loss tail.  systems/split_occ.py:163-215, systems/criterions.py:155-159
# --------------------------------------------------------------------------------------
def binary_cross_entropy(inp, target):
    return -(target * torch.log(inp) + (1 - target) * torch.log(1 - inp)).mean()


def loss_tail(out, batch, lambdas, sparsity_scale=1.0, has_mask=True, stage=0):
    """-> (loss, dict of unweighted terms), the reference's expressions one by one."""
    lam = lambda k: float(lambdas.get("lambda_" + k, 0.0))
    valid = out["rays_valid_full"][..., 0]
    t = {}
    t["rgb_mse"] = F.mse_loss(out["comp_rgb_full"][valid], batch["rgb"][valid])
    t["rgb_l1"] = F.l1_loss(out["comp_rgb_full"][valid], batch["rgb"][valid])
    loss = t["rgb_mse"] * lam("rgb_mse") + t["rgb_l1"] * lam("rgb_l1")
    if stage != 0:
        t["rgb_phys_mse"] = F.mse_loss(out["comp_rgb_phys_full"][valid], batch["rgb"][valid])
        t["rgb_phys_l1"] = F.l1_loss(out["comp_rgb_phys_full"][valid], batch["rgb"][valid])
        loss = loss + t["rgb_phys_mse"] * lam("rgb_phys_mse") + t["rgb_phys_l1"] * lam("rgb_phys_l1")
    t["eikonal"] = ((torch.linalg.norm(out["sdf_grad_samples"], ord=2, dim=-1) - 1.0) ** 2).mean()
    loss = loss + t["eikonal"] * lam("eikonal")
    opacity = torch.clamp(out["opacity"].squeeze(-1), 1.0e-3, 1.0 - 1.0e-3)
    t["mask"] = binary_cross_entropy(opacity, batch["fg_mask"].float())
    loss = loss + t["mask"] * (lam("mask") if has_mask else 0.0)
    t["opaque"] = binary_cross_entropy(opacity, opacity)
    loss = loss + t["opaque"] * lam("opaque")
    t["sparsity"] = torch.exp(-sparsity_scale * out["sdf_samples"].abs()).mean()
    loss = loss + t["sparsity"] * lam("sparsity")
    if lam("curvature") > 0:
        t["curvature"] = out["sdf_laplace_samples"].abs().mean()
        loss = loss + t["curvature"] * lam("curvature")
    return loss, t


# --------------------------------------------------------------------------------------
# checker-side plumbing: a model under test -> the plain structures the functions above take
# --------------------------------------------------------------------------------------
def params_from_model(model):
    """(meta, table, mlp, variance) of a NeuS-shaped model (``geometry.encoding`` = CompositeEncoding ->
    ProgressiveBandHashGrid -> tcnn.Encoding, ``geometry.network`` = VanillaMLP with weight_norm, ``variance``), as detached
    CPU leaves that require grad.  Used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
    geo = model.geometry
    enc = geo.encoding.encoding.encoding
    meta, n_params = grid_meta(enc.n_levels, enc.n_features_per_level, enc.log2_hashmap_size,
                               enc.base_resolution, enc.per_level_scale)
    table = enc.params.detach().cpu().clone().requires_grad_(True)
    mlp = []
    for m in geo.network.layers:
        if isinstance(m, torch.nn.Linear):
            mlp.append({"g": m.weight_g.detach().cpu().clone().requires_grad_(True),
                        "v": m.weight_v.detach().cpu().clone().requires_grad_(True),
                        "b": m.bias.detach().cpu().clone().requires_grad_(True)})
    var = model.variance.variance.detach().cpu().clone().requires_grad_(True)
    return meta, table, mlp, var
