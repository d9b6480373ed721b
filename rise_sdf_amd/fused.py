"""Fused SDF field for the finite-difference stencil: hash-grid encode -> 2-hidden-layer SDF MLP in
one autograd node, so that the encoding gradient goes straight from the MLP backward kernel into the
stencil-merging hash-grid scatter (no 35-wide autograd buffers, no per-layer activation round trips).

Forward:  x7 [S,7,3] -> enc [7S, 3+L*F] (rsdf_hashgrid_fwd) -> sdf7 [7S] (+ feature [S,N2] of the
          centre rows) (rsdf_sdfmlp_fd7_fwd).
Backward: d_sdf7 -> rsdf_sdfmlp_fd7_bwd (recomputes the hidden layers; weight/bias gradients and the
          encoding-gradient window [7S, L*F]) -> rsdf_hashgrid_bwd_fd7 -> d_table.

Mirrors what VolumeSDF.forward does for finite-difference normals (models/geometry.py:206-244) between
``points_d`` and ``points_d_sdf``; the reference runs it as 7 separate encode + MLP passes.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib as L
from ._lib import check, lib, ptr, require_device, stream_ptr


def supported(K0: int, H: int, N2: int, n_hidden_layers: int, hidden_act: str, out_act: str,
              n_features: int) -> bool:
    return (n_hidden_layers == 2 and hidden_act == "softplus100" and out_act == "none"
            and n_features == 2 and bool(lib().rsdf_sdfmlp_fd7_supported(int(K0), int(H), int(N2))))


class _SdfFieldFD7(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x7, table, w0, b0, w1, b1, w2, b2, meta, n_active, xyz_scale, xyz_offset,
                eps_unit, want_feature):
        xf = x7.detach().to(torch.float32).contiguous().view(-1, 3)
        tb = table.detach()
        ws = [t.detach().to(torch.float32).contiguous() for t in (w0, b0, w1, b1, w2, b2)]
        require_device(xf, tb, *ws)
        n_rows = xf.shape[0]
        assert n_rows % 7 == 0, "x7 must be [S,7,3]"
        S = n_rows // 7
        LF = meta.n_levels * meta.n_features
        K0 = 3 + LF
        H, N2 = ws[0].shape[0], ws[4].shape[0]
        dev = xf.device
        st = stream_ptr()
        enc = torch.empty(n_rows, K0, dtype=torch.float32, device=dev)
        check(lib().rsdf_hashgrid_fwd(ptr(xf), ptr(tb), ctypes.byref(meta), n_rows, n_active, ptr(enc),
                                      K0, 3, 1, float(xyz_scale), float(xyz_offset), st), "hashgrid_fwd")
        sdf7 = torch.empty(n_rows, dtype=torch.float32, device=dev)
        feature = torch.empty(S, N2, dtype=torch.float32, device=dev) if want_feature else None
        check(lib().rsdf_sdfmlp_fd7_fwd(ptr(enc), K0, K0, H, N2, *[ptr(t) for t in ws], S, ptr(sdf7),
                                        ptr(feature), st), "sdfmlp_fd7_fwd")
        ctx.save_for_backward(xf, enc, *ws)
        ctx.meta, ctx.n_active, ctx.eps_unit, ctx.n_params = meta, n_active, float(eps_unit), tb.numel()
        ctx.dims = (S, K0, H, N2, LF)
        ctx.set_materialize_grads(False)
        if feature is None:
            return sdf7, None
        return sdf7, feature

    @staticmethod
    def backward(ctx, g_sdf7, g_feature):
        if g_feature is not None:
            raise L.RiseSdfHipError(
                "the fused stencil field only back-propagates through the SDF column; use "
                "VolumeSDF.forward (per-layer kernels) when the feature vector needs a gradient")
        xf, enc, w0, b0, w1, b1, w2, b2 = ctx.saved_tensors
        S, K0, H, N2, LF = ctx.dims
        dev = xf.device
        st = stream_ptr()
        if g_sdf7 is None:
            return (None,) * 14
        g = g_sdf7.detach().to(torch.float32).contiguous()
        need_table = ctx.needs_input_grad[1]
        d_enc = torch.empty(7 * S, LF, dtype=torch.float32, device=dev) if need_table else None
        dw0, db0 = torch.zeros_like(w0), torch.zeros_like(b0)
        dw1, db1 = torch.zeros_like(w1), torch.zeros_like(b1)
        dw2, db2 = torch.zeros_like(w2), torch.zeros_like(b2)
        check(lib().rsdf_sdfmlp_fd7_bwd(ptr(enc), K0, K0, H, N2, ptr(w0), ptr(b0), ptr(w1), ptr(b1),
                                        ptr(w2), ptr(b2), S, ptr(g), 3, LF, ptr(d_enc), LF, ptr(dw0),
                                        ptr(db0), ptr(dw1), ptr(db1), ptr(dw2), ptr(db2), st),
              "sdfmlp_fd7_bwd")
        dt = None
        if need_table:
            dt = torch.zeros(ctx.n_params, dtype=torch.float32, device=dev)
            nbytes = int(lib().rsdf_hashgrid_bwd_fd7_scratch_bytes(ctypes.byref(ctx.meta), S,
                                                                   ctx.n_active, ctx.eps_unit))
            if nbytes < 0:
                raise L.RiseSdfHipError("hashgrid_bwd_fd7: unsupported level layout")
            scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            check(lib().rsdf_hashgrid_bwd_fd7(ptr(xf), ptr(d_enc), ctypes.byref(ctx.meta), S, ctx.n_active,
                                              LF, 0, ctx.eps_unit, ptr(dt), ptr(scratch), nbytes, st),
                  "hashgrid_bwd_fd7")
        return (None, dt, dw0, db0, dw1, db1, dw2, db2, None, None, None, None, None, None)


def sdf_field_fd7(x7, table, weights, meta, n_active, xyz_scale, xyz_offset, eps_unit,
                  want_feature=False):
    """weights = [(w0,b0),(w1,b1),(w2,b2)] effective (already weight-normalised) layer parameters.
    Returns (sdf7 [7S], feature [S,N2] or None)."""
    (w0, b0), (w1, b1), (w2, b2) = weights
    return _SdfFieldFD7.apply(x7, table, w0, b0, w1, b1, w2, b2, meta, int(n_active), xyz_scale,
                              xyz_offset, eps_unit, bool(want_feature))
