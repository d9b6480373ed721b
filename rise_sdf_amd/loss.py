"""Loss tail of the training step on the path's outputs: the expressions of
``systems/split_occ.py:163-215`` (masked RGB MSE / L1 on ``comp_rgb_full`` and, at stage 1, ``comp_rgb_phys_full``;
eikonal; mask and "opaque" binary cross entropy on the clamped opacity, ``criterions.py:155-159``; sparsity;
curvature) evaluated by two reduction kernels, their gradients by two elementwise kernels.  No host reads: the
sums stay on the device and the backward coefficients are a device array.

Not included (they stay ordinary tensor ops in a training loop): the distortion loss (third-party
``torch_efficient_distloss``, lambda 0 in the shipped configs), the emitter distillation term and
``model.regularizations``.
"""
from __future__ import annotations

import torch

from ._lib import check, lib, ptr, require_device, stream_ptr


def _f(t):
    return None if t is None else t.detach().to(torch.float32).contiguous()


_LAMBDA_CACHE = {}


def _lambda_tensor(lam, dev):
    """The nine coefficients on the device.  They change rarely (a scheduled lambda at most once per step), so the
    host-to-device copy -- a synchronising call in the middle of the step -- is made once per distinct tuple."""
    key = (tuple(lam), str(dev))
    t = _LAMBDA_CACHE.get(key)
    if t is None:
        if len(_LAMBDA_CACHE) > 64:
            _LAMBDA_CACHE.clear()
        t = _LAMBDA_CACHE[key] = torch.tensor(lam, dtype=torch.float32, device=dev)
    return t


class _LossTail(torch.autograd.Function):
    @staticmethod
    def forward(ctx, comp_rgb, comp_rgb_phys, opacity, sdf, sdf_grad, laplace, target, rays_valid, fg_mask,
                lam, sparsity_scale):
        rgb, phys, op = _f(comp_rgb), _f(comp_rgb_phys), _f(opacity).reshape(-1)
        s, g, lp = _f(sdf).reshape(-1), _f(sdf_grad), _f(laplace)
        tg, fm = _f(target), (None if fg_mask is None else _f(fg_mask).reshape(-1))
        va = rays_valid.reshape(-1).contiguous()
        va = va.view(torch.uint8) if va.dtype == torch.bool else va.to(torch.uint8)
        require_device(rgb, phys, op, s, g, lp, tg, fm, va)
        dev, N, S = rgb.device, rgb.shape[0], s.shape[0]
        sums = torch.zeros(10, dtype=torch.float64, device=dev)
        st = stream_ptr()
        check(lib().rsdf_loss_rays_fwd(ptr(rgb), ptr(phys), ptr(tg), ptr(va), ptr(op), ptr(fm), N, ptr(sums), st),
              "loss_rays_fwd")
        check(lib().rsdf_loss_samples_fwd(ptr(s), ptr(g), ptr(lp), float(sparsity_scale), S, ptr(sums[7:]), st),
              "loss_samples_fwd")
        n3 = 3.0 * sums[4]
        terms = torch.stack([sums[0] / n3, sums[1] / n3, sums[2] / n3, sums[3] / n3, sums[5] / N, sums[6] / N,
                             sums[7] / max(S, 1), sums[8] / max(S, 1), sums[9] / max(S, 1)]).to(torch.float32)
        if S == 0:   # .mean() of an empty tensor
            terms[6:] = float("nan")
        if phys is None:
            terms[2:4] = 0.0
        if fm is None:
            terms[4] = 0.0
        if lp is None:
            terms[8] = 0.0
        lam_t = _lambda_tensor(lam, dev)
        # per-term normalisers for the backward: 1 / (3 valid) x4, 1 / N x2, 1 / S x3
        inv = torch.cat([(1.0 / n3).to(torch.float32).expand(4), torch.full((2,), 1.0 / N, device=dev),
                         torch.full((3,), 1.0 / max(S, 1), device=dev)])
        ctx.save_for_backward(rgb, tg, va, op, s, g, lam_t * inv,
                              *(t for t in (phys, fm, lp) if t is not None))
        ctx.has = (phys is not None, fm is not None, lp is not None)
        ctx.scale = float(sparsity_scale)
        ctx.shapes = (opacity.shape, sdf.shape)
        ctx.mark_non_differentiable(terms)
        return (terms * lam_t).sum(), terms

    @staticmethod
    def backward(ctx, g_total, g_terms):
        rgb, tg, va, op, s, g, w, *rest = ctx.saved_tensors
        rest = list(rest)
        phys = rest.pop(0) if ctx.has[0] else None
        fm = rest.pop(0) if ctx.has[1] else None
        lp = rest.pop(0) if ctx.has[2] else None
        # d loss / d term_k = g_total * lambda_k; the per-term values are returned for logging only
        coef = (w * g_total).to(torch.float32).contiguous()
        N, S = rgb.shape[0], s.shape[0]
        d_rgb, d_op = torch.empty_like(rgb), torch.empty_like(op)
        d_phys = torch.empty_like(phys) if phys is not None else None
        d_s, d_g = torch.empty_like(s), torch.empty_like(g)
        d_lp = torch.empty_like(lp) if lp is not None else None
        st = stream_ptr()
        check(lib().rsdf_loss_rays_bwd(ptr(rgb), ptr(phys), ptr(tg), ptr(va), ptr(op), ptr(fm), ptr(coef), N,
                                       ptr(d_rgb), ptr(d_phys), ptr(d_op), st), "loss_rays_bwd")
        check(lib().rsdf_loss_samples_bwd(ptr(s), ptr(g), ptr(lp), ctx.scale, ptr(coef[6:]), S, ptr(d_s), ptr(d_g),
                                          ptr(d_lp), st), "loss_samples_bwd")
        return (d_rgb, d_phys, d_op.view(ctx.shapes[0]), d_s.view(ctx.shapes[1]), d_g, d_lp, None, None, None,
                None, None)


TERMS = ("rgb_mse", "rgb_l1", "rgb_phys_mse", "rgb_phys_l1", "mask", "opaque", "eikonal", "sparsity", "curvature")


def loss_tail(out, batch, lambdas, sparsity_scale=1.0, has_mask=True):
    """``out``: the model's output dict; ``batch``: ``rgb`` [N,3], ``fg_mask`` [N]; ``lambdas``: dict with the
    reference's keys ``lambda_rgb_mse, lambda_rgb_l1, lambda_rgb_phys_mse, lambda_rgb_phys_l1, lambda_mask,
    lambda_opaque, lambda_eikonal, lambda_sparsity, lambda_curvature`` (missing = 0).
    -> (loss, {name: value}) with the reference's log names (``loss_<name>``).  The weighted sum is differentiable
    w.r.t. comp_rgb_full, comp_rgb_phys_full, opacity, sdf_samples, sdf_grad_samples, sdf_laplace_samples."""
    lam = [float(lambdas.get("lambda_" + k, 0.0)) for k in TERMS]
    if not has_mask:
        lam[4] = 0.0
    phys = out.get("comp_rgb_phys_full", None)
    if phys is None:
        lam[2] = lam[3] = 0.0
    lap = out.get("sdf_laplace_samples", None) if lam[8] > 0 else None
    total, terms = _LossTail.apply(out["comp_rgb_full"], phys, out["opacity"], out["sdf_samples"],
                                   out["sdf_grad_samples"], lap, batch["rgb"], out["rays_valid_full"],
                                   batch.get("fg_mask", None), tuple(lam), float(sparsity_scale))
    return total, {"loss_" + k: terms[i].detach() for i, k in enumerate(TERMS)}
