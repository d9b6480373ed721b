"""Fused SDF field for the finite-difference stencil: hash-grid encode -> 2-hidden-layer SDF MLP as
ONE autograd node over tap-major structure-of-arrays buffers, so that every kernel streams full
cache lines and nothing 35 columns wide ever goes through autograd.

  x7t    [7][S][3]      stencil points (rsdf_fd_points, tap_major)
  planes [L][7][S][2]   hash features per level / tap / sample

Forward:  x7t -> planes (rsdf_hashgrid_fwd_fd7: one gather of the centre cell + 4 corners per displaced
          tap) -> sdf7t [7][S] (+ feature [S,N2] of the centre taps) (rsdf_sdfmlp_fd7_fwd).
          Round 4, H = 64 in fp32: the "x2" form (csrc/mlp_x2.hip) -- every fp32 matrix operand as two fp16 parts, three
          matrix instructions per product instead of six, and the gather writes the MLP kernels' input image PRE-SPLIT
          (``x2`` [S / 32][7][2][36][32] fp16, the same bytes as the fp32 planes; rsdf_hashgrid_fwd_fd7_x2), which both
          MLP kernels consume as it is (rsdf_sdfmlp_fd7_fwd_x2 / _bwd_x2).  ``RSDF_X2=0`` keeps the round-3 kernels.
Backward: d_sdf7t (+ d_feature, first pushed through the feature rows of the last layer into a [S,H] scratch)
          -> rsdf_sdfmlp_fd7_bwd (recomputes the hidden layers; weight / bias gradients and d_planes)
          -> rsdf_hashgrid_bwd_fd7 (merge, bin through LDS, reduce in LDS) ->
          d_table; the feature rows of dW2 come from rsdf_linear_bwd_weight on the saved centre h2.

This is what VolumeSDF.forward does for finite-difference normals between ``points_d`` and
``points_d_sdf`` (models/geometry.py:229-244); the reference runs it as 7 encode + MLP passes.
"""
from __future__ import annotations

import ctypes
import os

import torch

from . import _lib as L
from ._lib import check, lib, ptr, require_device, stream_ptr


def supported(K0: int, H: int, N2: int, n_hidden_layers: int, hidden_act: str, out_act: str,
              n_features: int) -> bool:
    return (n_hidden_layers == 2 and hidden_act == "softplus100" and out_act == "none"
            and n_features == 2 and (K0 - 3) % 2 == 0 and (K0 - 3) // 2 <= 16
            and bool(lib().rsdf_sdfmlp_fd7_supported(int(K0), int(H), int(N2))))


def x2_parts(K0: int, H: int, N2: int, precision: str) -> int:
    """0: the round-3 kernels on fp32 planes.  2: the two-part fp16 form with the pre-split input image (H = 64 and 128 in
    fp32: forward + quad backward).  1: the same kernels with ONE fp16 part -- precision 'fp16', the 16-bit mode of this
    node (``precision: fp16`` in the SDF network's config; the per-layer kernels of such a network use the bf16 build)."""
    if H not in (64, 128) or not bool(lib().rsdf_sdfmlp_fd7_x2_supported(int(K0), int(H), int(N2))):
        return 0
    if precision == "fp16":
        return 1
    if (precision == "fp32" and os.environ.get("RSDF_X2", "1") != "0" and not L.range_free("x2")
            and os.environ.get("RSDF_MLP_FWD", "") != "coop"
            and os.environ.get("RSDF_MLP_BWD", "") == ""):
        return 2
    return 0


def x2_reroute_mode() -> int:
    """The x2 backward's range guard: 1 (default) = decided per launch on the device (a launch whose largest gradient row is
    an outlier relative to the bulk runs on the range-free kernels), ``RSDF_X2_REROUTE=0`` never, ``=force`` always."""
    v = os.environ.get("RSDF_X2_REROUTE", "1")
    return 0 if v == "0" else (2 if v == "force" else 1)


def use_x2(K0: int, H: int, N2: int, precision: str) -> bool:
    return x2_parts(K0, H, N2, precision) > 0


class _SdfFieldFD7(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x7t, table, w0, b0, w1, b1, w2, b2, meta, n_active, xyz_scale, xyz_offset,
                eps_unit, want_feature, points=None, radius=None, eps=None, precision="fp32"):
        xf = None if x7t is None else x7t.detach().to(torch.float32).contiguous()     # (None: positions only, the x2 form)
        assert xf is None or (xf.dim() == 3 and xf.shape[0] == 7 and xf.shape[2] == 3), "x7t must be [7,S,3]"
        assert xf is not None or points is not None, "x7t or points"
        tb = table.detach()
        ws = [t.detach().to(torch.float32).contiguous() for t in (w0, b0, w1, b1, w2, b2)]
        require_device(xf, tb, *ws)
        S = xf.shape[1] if xf is not None else points.shape[0]
        Lv = int(meta.n_levels)
        H, N2 = ws[0].shape[0], ws[4].shape[0]
        dev = tb.device
        st = stream_ptr()
        x2 = None
        pts = None
        if points is not None:
            pts = points.detach().to(torch.float32).contiguous()
            require_device(pts)
            assert pts.shape == (S, 3), "points must be [S,3]"
        parts = x2_parts(3 + 2 * Lv, H, N2, precision) if S > 0 else 0
        if precision == "fp16" and S > 0 and not parts:
            raise L.RiseSdfHipError("precision 'fp16' of the fused SDF field needs H = 64 or 128")
        if xf is None and S > 0 and not parts:
            raise L.RiseSdfHipError("the fused SDF field without x7t needs the x2 kernels (H = 64 / 128, RSDF_X2 unset)")
        if parts:
            if os.environ.get("RSDF_CHECK", "0") == "1":
                # the x2 form's fp16 range preconditions (csrc/mlp_x2.hip): a violation would show as inf / nan outputs
                wmax = max(float(t.abs().max()) for t in ws)
                if not (wmax < 1023.0 and float(tb.abs().max()) < 255.0):
                    raise L.RiseSdfHipError(f"RSDF_CHECK x2: |weight| {wmax:.3g} must be < 1023 and |table| "
                                            f"{float(tb.abs().max()):.3g} < 255 (fp16 class scales); use RSDF_X2=0")
            x2 = torch.empty(int(lib().rsdf_x2_bytes(S, parts)), dtype=torch.uint8, device=dev)
            check(lib().rsdf_hashgrid_fwd_fd7_x2(None if pts is not None else ptr(xf), ptr(pts), float(radius or 0.0),
                                                 float(eps or 0.0), ptr(tb), ctypes.byref(meta), S, n_active,
                                                 float(xyz_scale), float(xyz_offset), parts, ptr(x2), st),
                  "hashgrid_fwd_fd7_x2")
            planes = x2
        else:
            planes = torch.empty(Lv, 7, S, 2, dtype=torch.float32, device=dev)
        if x2 is not None:
            pass
        elif points is not None:
            # the hash kernels derive the stencil from the world-space centres (12 instead of 84 bytes per sample and
            # level); x7t still feeds the xyz columns of the MLP kernels, which read it once
            check(lib().rsdf_hashgrid_fwd_fd7_pts(ptr(pts), float(radius), float(eps), ptr(tb), ctypes.byref(meta), S,
                                                  n_active, ptr(planes), st), "hashgrid_fwd_fd7_pts")
        else:
            check(lib().rsdf_hashgrid_fwd_fd7(ptr(xf), ptr(tb), ctypes.byref(meta), S, n_active, ptr(planes),
                                              st), "hashgrid_fwd_fd7")
        sdf7t = torch.empty(7, S, dtype=torch.float32, device=dev)
        feature = torch.empty(S, N2, dtype=torch.float32, device=dev) if want_feature else None
        h2c = torch.empty(S, H, dtype=torch.float32, device=dev) if want_feature else None
        if x2 is not None:
            # (range guard: a non-finite output is counted in the device's status words; _lib.poll_status raises at the
            # next host read, naming RSDF_X2=0)
            check(lib().rsdf_sdfmlp_fd7_fwd_x2(ptr(x2), parts, Lv, H, N2, *[ptr(t) for t in ws], S, ptr(sdf7t), ptr(feature),
                                               ptr(h2c), ptr(L.status(dev)), st), "sdfmlp_fd7_fwd_x2")
        else:
            if L._DEBUG_SYNC:           # (debug aid: the arguments of the launch that a fault is about to be pinned on)
                with open(L._DEBUG_SYNC, "a") as f:
                    f.write(f"  sdfmlp_fd7_fwd args: S={S} Lv={Lv} n_active={n_active} H={H} N2={N2} feature={want_feature} "
                            f"xf={None if xf is None else (tuple(xf.shape), hex(xf.data_ptr()))} planes={tuple(planes.shape)}@{planes.data_ptr():#x} "
                            f"pts={None if pts is None else tuple(pts.shape)} ws={[tuple(t.shape) for t in ws]} "
                            f"ws_ptr={[hex(t.data_ptr()) for t in ws]}\n")
            check(L.mlp_fn("rsdf_sdfmlp_fd7_fwd", precision)(ptr(xf), ptr(planes), Lv, n_active, float(xyz_scale),
                                            float(xyz_offset), H, N2, *[ptr(t) for t in ws], S, ptr(sdf7t),
                                            ptr(feature), ptr(h2c), st), "sdfmlp_fd7_fwd")
        ctx.x2 = parts
        ctx.save_for_backward(xf, planes, *ws)
        ctx.h2c = h2c
        ctx.pts, ctx.radius, ctx.eps, ctx.precision = pts, radius, eps, precision
        ctx.meta, ctx.n_active, ctx.eps_unit, ctx.n_params = meta, n_active, float(eps_unit), tb.numel()
        ctx.xyz = (float(xyz_scale), float(xyz_offset))
        ctx.dims = (S, Lv, H, N2)
        ctx.set_materialize_grads(False)
        return sdf7t, feature

    @staticmethod
    def backward(ctx, g_sdf7t, g_feature):
        if g_sdf7t is None and g_feature is None:
            return (None,) * 18
        xf, planes, w0, b0, w1, b1, w2, b2 = ctx.saved_tensors
        S, Lv, H, N2 = ctx.dims
        dev = planes.device
        st = stream_ptr()
        g = torch.zeros(7, S, dtype=torch.float32, device=dev) if g_sdf7t is None \
            else g_sdf7t.detach().to(torch.float32).contiguous()
        gf = None if g_feature is None else g_feature.detach().to(torch.float32).contiguous()
        need_table = ctx.needs_input_grad[1]
        # (backward-transient buffers come from the per-stream arena, _lib.workspace: sizes follow the chunk's sample count)
        d_planes = L.workspace_f32("fd7.d_planes", (Lv, 7, S, 2), dev) if need_table else None
        dw0, db0 = torch.zeros_like(w0), torch.zeros_like(b0)
        dw1, db1 = torch.zeros_like(w1), torch.zeros_like(b1)
        dw2, db2 = torch.zeros_like(w2), torch.zeros_like(b2)
        dh2c = L.workspace_f32("fd7.dh2c", (S, H), dev) if gf is not None else None
        if ctx.x2:
            guard = torch.empty(8, dtype=torch.int32, device=dev)
            # the backward's range guard (csrc/mlp_x2.hip): a launch whose largest gradient row is an outlier relative to the
            # bulk runs on the range-free kernels instead, decided on the device; that route rebuilds x7t here and the planes
            # in place of d_planes
            reroute = x2_reroute_mode() if (need_table and ctx.x2 == 2) else 0
            x7s = L.workspace_f32("fd7.x7t_reroute", (7, S, 3), dev) if reroute else None
            check(lib().rsdf_sdfmlp_fd7_bwd_x2(ptr(planes), ctx.x2, Lv, ctx.n_active, H, N2, ptr(w0), ptr(b0), ptr(w1), ptr(b1),
                                               ptr(w2), ptr(b2), S, ptr(g), ptr(gf), ptr(dh2c), ptr(guard), ptr(x7s), reroute,
                                               ptr(d_planes), ptr(dw0), ptr(db0), ptr(dw1), ptr(db1), ptr(dw2), ptr(db2),
                                               ptr(L.status(dev)), st), "sdfmlp_fd7_bwd_x2")
        else:
            check(L.mlp_fn("rsdf_sdfmlp_fd7_bwd", ctx.precision)(ptr(xf), ptr(planes), Lv, ctx.n_active, ctx.xyz[0], ctx.xyz[1],
                                            H, N2, ptr(w0), ptr(b0), ptr(w1), ptr(b1), ptr(w2), ptr(b2), S,
                                            ptr(g), ptr(gf), ptr(dh2c), ptr(d_planes), ptr(dw0), ptr(db0), ptr(dw1),
                                            ptr(db1), ptr(dw2), ptr(db2), st), "sdfmlp_fd7_bwd")
        if gf is not None:
            # feature rows of the last layer: dW2 += d_feature^T h2(centre), db2 += colsum(d_feature)
            check(L.mlp_fn("rsdf_linear_bwd_weight", "fp32" if ctx.x2 else ctx.precision)(ptr(gf), N2, ptr(ctx.h2c), H, S, H, N2, ptr(dw2),
                                                                    ptr(db2), st),
                  "linear_bwd_weight (feature rows)")
        dt = None
        if need_table:
            dt = torch.zeros(ctx.n_params, dtype=torch.float32, device=dev)
            nbytes = int(lib().rsdf_hashgrid_bwd_fd7_scratch_bytes(ctypes.byref(ctx.meta), S,
                                                                   ctx.n_active, ctx.eps_unit))
            if nbytes < 0:
                raise L.RiseSdfHipError("hashgrid_bwd_fd7: unsupported level layout")
            scratch = L.workspace("fd7.queues", nbytes, dev)
            if ctx.pts is not None:
                check(lib().rsdf_hashgrid_bwd_fd7_pts(ptr(ctx.pts), float(ctx.radius), float(ctx.eps), ptr(d_planes),
                                                      ctypes.byref(ctx.meta), S, ctx.n_active, ctx.eps_unit, ptr(dt),
                                                      ptr(scratch), nbytes, st), "hashgrid_bwd_fd7_pts")
            else:
                check(lib().rsdf_hashgrid_bwd_fd7(ptr(xf), ptr(d_planes), ctypes.byref(ctx.meta), S,
                                                  ctx.n_active, ctx.eps_unit, ptr(dt), ptr(scratch), nbytes,
                                                  st), "hashgrid_bwd_fd7")
        return (None, dt, dw0, db0, dw1, db1, dw2, db2, None, None, None, None, None, None, None, None, None, None)


def sdf_field_fd7(x7t, table, weights, meta, n_active, xyz_scale, xyz_offset, eps_unit,
                  want_feature=False, points=None, radius=None, eps=None, precision="fp32"):
    """weights = [(w0,b0),(w1,b1),(w2,b2)] effective (already weight-normalised) layer parameters.
    ``points`` [S,3] (world-space sample centres, with ``radius`` and the FD ``eps``): the hash kernels derive the
    stencil from them instead of re-reading x7t once per level.
    Returns (sdf7t [7,S], feature [S,N2] or None)."""
    (w0, b0), (w1, b1), (w2, b2) = weights
    if points is not None:
        assert radius is not None and eps is not None
    return _SdfFieldFD7.apply(x7t, table, w0, b0, w1, b1, w2, b2, meta, int(n_active), xyz_scale,
                              xyz_offset, eps_unit, bool(want_feature), points, radius, eps, precision)
