"""torch.autograd wrappers over the C ABI.  PyTorch supplies device memory, the current stream and
autograd bookkeeping; every computation is a HIP kernel in librisesdf_hip.so.

Reference interfaces mirrored (paths relative to the upstream RISE-SDF tree):
  nerfacc 0.5.3 call sites ........ models/volrend.py:851-885, models/split_mixed_occ.py:200-208,264-272
  vendored nerfacc 0.3.5 .......... lib/nerfacc/{intersection,ray_marching,pack,vol_rendering}.py
  tcnn.Encoding ................... models/network_utils.py:47-50,59
  VanillaMLP ...................... models/network_utils.py:109-157
  get_alpha / VolumeSDF FD ........ models/split_mixed_occ.py:151-177, models/geometry.py:229-244
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

from . import _lib as L
from ._lib import check, lib, ptr, require_device, stream_ptr


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()


def _u8(t: torch.Tensor) -> torch.Tensor:
    """bool / uint8 grid or mask as a contiguous uint8 view (no copy for bool)."""
    t = t.contiguous()
    return t.view(torch.uint8) if t.dtype == torch.bool else t.to(torch.uint8)


# ------------------------------------------------------------------------------------------------
# M1
# ------------------------------------------------------------------------------------------------
@torch.no_grad()
def ray_aabb_intersect(rays_o, rays_d, aabb):
    """lib/nerfacc/intersection.py:13-50 -> (t_min, t_max), both [N]."""
    o, d, a = _f32c(rays_o), _f32c(rays_d), _f32c(aabb)
    require_device(o, d, a)
    n = o.shape[0]
    t_min = torch.empty(n, dtype=torch.float32, device=o.device)
    t_max = torch.empty(n, dtype=torch.float32, device=o.device)
    check(lib().rsdf_ray_aabb_intersect(ptr(o), ptr(d), ptr(a), n, ptr(t_min), ptr(t_max),
                                        stream_ptr()), "ray_aabb_intersect")
    return t_min, t_max


# ------------------------------------------------------------------------------------------------
# M3 / M4
# ------------------------------------------------------------------------------------------------
def _scan_scratch(n, device):
    return torch.empty(max(int(lib().rsdf_scan_scratch_bytes(n)), 4), dtype=torch.uint8, device=device)


# The marcher knows the packed_info of the ray_indices it returns; nerfacc's sampling() API hands on ray_indices only
# and the renderer packs them again (lib/nerfacc/pack.py:47-78: a run-boundary pass over all samples + a scan).  The last
# marcher result is remembered by the IDENTITY of its ray_indices tensor (a weak reference: no address or shape
# comparison that a recycled allocation could satisfy); pack_info() returns it for that very tensor.
# Thread-local (a second host thread -- a data-parallel replica, a prefetching evaluator -- has its own marcher calls), and the
# sample count is part of the key.
import threading as _threading

_LAST_PACKED_TLS = _threading.local()


def _last_packed():
    lp = getattr(_LAST_PACKED_TLS, "v", None)
    if lp is None:
        lp = _LAST_PACKED_TLS.v = [None, None, 0, 0, None]   # weakref to ray_indices, packed_info, n_rays, n_samples, stream
    return lp


def _remember_packed(ray_indices, packed, n_rays):
    import weakref
    lp = _last_packed()
    lp[0], lp[1], lp[2], lp[3] = weakref.ref(ray_indices), packed, int(n_rays), int(ray_indices.numel())
    lp[4] = torch.cuda.current_stream(packed.device) if packed.is_cuda else None


_STAGE_BYTES_MAX = 2 << 30      # staging slots beyond this (n_rays x stride x 8 bytes) are not worth their footprint


def _march_stage(n, t_range_hint, step_size, dev):
    """-> (stride, t0 slots, t1 slots) for the staged marcher, or (0, None, None): the per-ray slot holds every step a ray
    can take over ``t_range_hint`` (the caller's bound on t_max - t_min: the ROI's diagonal for unit directions); a ray that
    takes more is marched twice as before, so the hint is a performance parameter, not a correctness one.  RSDF_MARCH=two_pass
    keeps the reference's two marching passes (A/B, tests)."""
    if t_range_hint is None or n == 0 or os.environ.get("RSDF_MARCH") == "two_pass":
        return 0, None, None
    stride = int(float(t_range_hint) / float(step_size)) + 4
    if stride < 1 or n * stride * 8 > _STAGE_BYTES_MAX:
        return 0, None, None
    buf = L.workspace("march.stage", n * stride * 8, dev).view(torch.float32)
    return stride, buf[:n * stride], buf[n * stride:]


def _march_count(o, d, tn, tf, r, b, step_size, cone_angle, counts, stage, st):
    rx, ry, rz = b.shape
    n = o.shape[0]
    if stage[0]:
        check(lib().rsdf_march_count_staged(ptr(o), ptr(d), ptr(tn), ptr(tf), ptr(r), ptr(b), rx, ry, rz, float(step_size),
                                            float(cone_angle), n, ptr(counts), stage[0], ptr(stage[1]), ptr(stage[2]), st),
              "march_count_staged")
    else:
        check(lib().rsdf_march_count(ptr(o), ptr(d), ptr(tn), ptr(tf), ptr(r), ptr(b), rx, ry, rz, float(step_size),
                                     float(cone_angle), n, ptr(counts), st), "march_count")


def _march_write(o, d, tn, tf, r, b, step_size, cone_angle, packed, counts, stage, ri, ts, te, st):
    rx, ry, rz = b.shape
    n = o.shape[0]
    if stage[0]:
        check(lib().rsdf_march_write_staged(ptr(o), ptr(d), ptr(tn), ptr(tf), ptr(r), ptr(b), rx, ry, rz, float(step_size),
                                            float(cone_angle), n, ptr(packed), ptr(counts), stage[0], ptr(stage[1]),
                                            ptr(stage[2]), ptr(ri), ptr(ts), ptr(te), st), "march_write_staged")
    else:
        check(lib().rsdf_march_write(ptr(o), ptr(d), ptr(tn), ptr(tf), ptr(r), ptr(b), rx, ry, rz, float(step_size),
                                     float(cone_angle), n, ptr(packed), ptr(ri), ptr(ts), ptr(te), st), "march_write")


@torch.no_grad()
def march(rays_o, rays_d, t_min, t_max, roi, binary, step_size, cone_angle=0.0, t_range_hint=None):
    """The two-pass marcher of lib/nerfacc/cuda/csrc/ray_marching.cu:194-289.
    Returns packed_info int32 [N,2], ray_indices int64 [S], t_starts [S], t_ends [S].
    One host read-back of the sample total, as in the reference (:261).  ``t_range_hint`` (a host float >= t_max - t_min of
    the typical ray): the rays are marched ONCE, the count pass parks their samples for the write pass (_march_stage)."""
    o, d, tn, tf, r = _f32c(rays_o), _f32c(rays_d), _f32c(t_min), _f32c(t_max), _f32c(roi)
    b = _u8(binary)
    require_device(o, d, tn, tf, r, b)
    assert b.dim() == 3, "grid_binary must be [res_x, res_y, res_z]"
    n = o.shape[0]
    dev = o.device
    counts = torch.empty(n, dtype=torch.int32, device=dev)
    packed = torch.empty(n, 2, dtype=torch.int32, device=dev)
    total = torch.zeros(1, dtype=torch.int32, device=dev)
    st = stream_ptr()
    stage = _march_stage(n, t_range_hint, step_size, dev)
    _march_count(o, d, tn, tf, r, b, step_size, cone_angle, counts, stage, st)
    scratch = _scan_scratch(n, dev)
    check(lib().rsdf_pack_from_counts(ptr(counts), n, ptr(packed), ptr(total), ptr(scratch), st),
          "pack_from_counts")
    S = int(total.item())
    L.poll_status(dev)                 # (the stream is drained: the kernels' sticky status words ride along with this read)
    ri = torch.empty(S, dtype=torch.int64, device=dev)
    ts = torch.empty(S, dtype=torch.float32, device=dev)
    te = torch.empty(S, dtype=torch.float32, device=dev)
    if S > 0:
        _march_write(o, d, tn, tf, r, b, step_size, cone_angle, packed, counts, stage, ri, ts, te, st)
    _remember_packed(ri, packed, n)
    return packed, ri, ts, te


@torch.no_grad()
def march_capped(rays_o, rays_d, t_min, t_max, roi, binary, step_size, capacity, cone_angle=0.0, t_range_hint=None):
    """The marcher WITHOUT its host read: count -> device scan -> write into buffers of ``capacity`` samples that the
    caller sized from an earlier call.  packed_info is clamped to the capacity on the device (a ray whose samples would
    not fit keeps the ones that do), the tail of the buffers is a harmless dummy sample (ray 0, t = 0) that no ray's
    packed_info covers.  -> (packed_info, ray_indices [cap], t_starts [cap], t_ends [cap], total int32 [1] on the
    device: the TRUE sample count; total > capacity means the set was truncated and must be redone)."""
    o, d, tn, tf, r = _f32c(rays_o), _f32c(rays_d), _f32c(t_min), _f32c(t_max), _f32c(roi)
    b = _u8(binary)
    require_device(o, d, tn, tf, r, b)
    n, dev = o.shape[0], o.device
    counts = torch.empty(n, dtype=torch.int32, device=dev)
    packed = torch.empty(n, 2, dtype=torch.int32, device=dev)
    total = torch.zeros(1, dtype=torch.int32, device=dev)
    st = stream_ptr()
    stage = _march_stage(n, t_range_hint, step_size, dev)
    _march_count(o, d, tn, tf, r, b, step_size, cone_angle, counts, stage, st)
    scratch = _scan_scratch(n, dev)
    check(lib().rsdf_pack_from_counts(ptr(counts), n, ptr(packed), ptr(total), ptr(scratch), st),
          "pack_from_counts")
    cap = int(capacity)
    packed[:, 1] = torch.minimum(packed[:, 1], (cap - packed[:, 0]).clamp_(min=0))
    packed[:, 0].clamp_(max=cap)        # (rays past the capacity: empty AND inside the buffers -- no kernel dereferences the offset
                                        #  of an empty ray, but a packed_info should never point outside its arrays: RSDF_CHECK)
    ri = torch.zeros(cap, dtype=torch.int64, device=dev)
    ts = torch.zeros(cap, dtype=torch.float32, device=dev)
    te = torch.zeros(cap, dtype=torch.float32, device=dev)
    if cap > 0:
        _march_write(o, d, tn, tf, r, b, step_size, cone_angle, packed, counts, stage, ri, ts, te, st)
    return packed, ri, ts, te, total


@torch.no_grad()
def query_occ(samples, roi, binary, return_cell=False):
    """lib/nerfacc/grid.py query_grid / ray_marching.cu:295-358 (AABB)."""
    x, r = _f32c(samples), _f32c(roi)
    b = _u8(binary)
    require_device(x, r, b)
    n = x.shape[0]
    occ = torch.empty(n, dtype=torch.uint8, device=x.device)
    cell = torch.empty(n, dtype=torch.int32, device=x.device) if return_cell else None
    check(lib().rsdf_query_occ(ptr(x), ptr(r), ptr(b), b.shape[0], b.shape[1], b.shape[2], n,
                               ptr(occ), ptr(cell), stream_ptr()), "query_occ")
    return (occ.bool(), cell) if return_cell else occ.bool()


# ------------------------------------------------------------------------------------------------
# M6 / M5
# ------------------------------------------------------------------------------------------------
@torch.no_grad()
def pack_info(ray_indices, n_rays):
    """lib/nerfacc/pack.py:47-78: sorted int64 ray_indices [S] -> packed_info int32 [n_rays, 2]."""
    lp = _last_packed()
    ref = lp[0]
    if (ref is not None and ref() is ray_indices and lp[2] == int(n_rays) and lp[3] == int(ray_indices.numel())
            and ray_indices._version == 0):
        # the marcher's own packed_info of this very tensor.  A consumer on ANOTHER stream (a sampling pass prefetched on a side
        # stream, bench.py) must tell the allocator: the cache entry is the tensor's only owner, and once the next marcher call
        # replaces it the block would be handed out again on the producing stream while this stream's kernels still read it
        # (a GPU memory fault in a prefetching bench run: garbage offsets)
        if lp[4] is not None:
            cur = torch.cuda.current_stream(lp[1].device)
            if cur != lp[4]:
                lp[1].record_stream(cur)
        return lp[1]
    ri = ray_indices.to(torch.int64).contiguous()
    require_device(ri)
    dev = ri.device
    counts = torch.empty(n_rays, dtype=torch.int32, device=dev)
    packed = torch.empty(n_rays, 2, dtype=torch.int32, device=dev)
    total = torch.empty(1, dtype=torch.int32, device=dev)
    st = stream_ptr()
    check(lib().rsdf_counts_from_ray_indices(ptr(ri), ri.numel(), n_rays, ptr(counts), st),
          "counts_from_ray_indices")
    scratch = _scan_scratch(n_rays, dev)
    check(lib().rsdf_pack_from_counts(ptr(counts), n_rays, ptr(packed), ptr(total), ptr(scratch), st),
          "pack_from_counts")
    return packed


@torch.no_grad()
def unpack_info(packed_info, n_samples):
    """lib/nerfacc/pack.py unpack_info / pack.cu:7-28."""
    pk = packed_info.to(torch.int32).contiguous()
    require_device(pk)
    ri = torch.empty(n_samples, dtype=torch.int64, device=pk.device)
    check(lib().rsdf_unpack_info(ptr(pk), pk.shape[0], ptr(ri), stream_ptr()), "unpack_info")
    return ri


@torch.no_grad()
def compact_samples(keep, ray_indices, t_starts, t_ends, count_out=None, fill_ray=None, extra=None):
    """Boolean-mask compaction of lib/nerfacc/ray_marching.py:213-218 (one host read of the count; with ``count_out`` (a
    list) the device count is appended to it instead and the un-sliced outputs are returned).  ``extra`` (float [S]): one
    more per-sample array compacted alongside, returned as a fourth output."""
    k = keep.contiguous().view(torch.uint8) if keep.dtype == torch.bool else keep.contiguous()
    ri, ts, te = ray_indices.contiguous(), _f32c(t_starts), _f32c(t_ends)
    require_device(k, ri, ts, te)
    n, dev = ri.numel(), ri.device
    off = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    ex = None if extra is None else _f32c(extra).reshape(-1)
    if fill_ray is None:
        ri_o, ts_o, te_o = torch.empty_like(ri), torch.empty_like(ts), torch.empty_like(te)
        ex_o = None if ex is None else torch.empty_like(ex)
    else:   # entries past the (device-side) count read as empty samples of ray ``fill_ray`` (a phantom last ray)
        ri_o, ts_o, te_o = torch.full_like(ri, int(fill_ray)), torch.zeros_like(ts), torch.zeros_like(te)
        ex_o = None if ex is None else torch.zeros_like(ex)
    scratch = _scan_scratch(n, dev)          # (a named local: alive until the kernels that use it are enqueued)
    check(lib().rsdf_compact_samples(ptr(k), ptr(ri), ptr(ts), ptr(te), n, ptr(off), ptr(cnt),
                                     ptr(scratch), ptr(ri_o), ptr(ts_o), ptr(te_o), ptr(ex), ptr(ex_o),
                                     stream_ptr()), "compact_samples")
    if count_out is not None:          # capacity mode: the caller reads this count together with the marcher's total
        count_out.append(cnt)
        return (ri_o, ts_o, te_o) if ex is None else (ri_o, ts_o, te_o, ex_o)
    m = int(cnt.item())
    L.poll_status(ri.device)
    return (ri_o[:m], ts_o[:m], te_o[:m]) if ex is None else (ri_o[:m], ts_o[:m], te_o[:m], ex_o[:m])


# ------------------------------------------------------------------------------------------------
# C1
# ------------------------------------------------------------------------------------------------
class _WeightFromAlpha(torch.autograd.Function):
    @staticmethod
    def forward(ctx, packed_info, alphas):
        a = _f32c(alphas)
        require_device(packed_info, a)
        w, t = torch.empty_like(a), torch.empty_like(a)
        check(lib().rsdf_weight_from_alpha_fwd(ptr(packed_info), ptr(a), packed_info.shape[0],
                                               ptr(w), ptr(t), stream_ptr()), "weight_from_alpha_fwd")
        ctx.save_for_backward(packed_info, a, w, t)
        ctx.mark_non_differentiable(t)
        return w, t

    @staticmethod
    def backward(ctx, gw, _gt):
        packed_info, a, w, t = ctx.saved_tensors
        gw = _f32c(gw)
        ga = torch.empty_like(a)
        # RSDF_C1_FMAD=0: the source's operations one rounding each instead of the reference binary's contraction (A/B, tests)
        check(lib().rsdf_weight_from_alpha_bwd_seq(ptr(packed_info), ptr(a), ptr(w), ptr(gw), packed_info.shape[0],
                                                   0 if os.environ.get("RSDF_C1_FMAD", "1") == "0" else 1, ptr(ga),
                                                   stream_ptr()), "weight_from_alpha_bwd")
        return None, ga


class _TransFromAlpha(torch.autograd.Function):
    @staticmethod
    def forward(ctx, packed_info, alphas):
        a = _f32c(alphas)
        require_device(packed_info, a)
        w, t = torch.empty_like(a), torch.empty_like(a)
        check(lib().rsdf_weight_from_alpha_fwd(ptr(packed_info), ptr(a), packed_info.shape[0],
                                               ptr(w), ptr(t), stream_ptr()), "weight_from_alpha_fwd")
        ctx.save_for_backward(packed_info, a, t)
        return t

    @staticmethod
    def backward(ctx, gt):
        packed_info, a, t = ctx.saved_tensors
        gt = _f32c(gt)
        ga = torch.empty_like(a)
        check(lib().rsdf_transmittance_from_alpha_bwd(ptr(packed_info), ptr(a), ptr(t), ptr(gt),
                                                      packed_info.shape[0], ptr(ga), stream_ptr()),
              "transmittance_from_alpha_bwd")
        return None, ga


CHECK = os.environ.get("RSDF_CHECK", "0") == "1"


def check_packed(packed_info, n_samples, what):
    """RSDF_CHECK=1 (debug; one host sync per call): every ray's [offset, offset + count) must lie inside the sample arrays
    it is used with.  The per-ray kernels trust packed_info; an inconsistent one (a stale cached pack, an under-sized
    capacity buffer) would read or write out of bounds."""
    if not CHECK or packed_info is None or packed_info.numel() == 0:
        return
    off, cnt = packed_info[:, 0].long(), packed_info[:, 1].long()
    bad = (off < 0) | (cnt < 0) | (off + cnt > int(n_samples))
    if bool(bad.any()):
        i = int(bad.nonzero()[0])
        raise RuntimeError(f"RSDF_CHECK {what}: ray {i} covers samples [{int(off[i])}, {int(off[i] + cnt[i])}) of {int(n_samples)}")


def _packed(ray_indices, packed_info, n_rays, n_samples=None, what="packed_info"):
    if packed_info is None:
        assert ray_indices is not None and n_rays is not None, \
            "either packed_info or (ray_indices, n_rays) is required"
        if CHECK and ray_indices.numel():
            lo, hi = int(ray_indices.min()), int(ray_indices.max())
            if lo < 0 or hi >= int(n_rays):
                raise RuntimeError(f"RSDF_CHECK {what}: ray_indices span [{lo}, {hi}] for {int(n_rays)} rays")
        packed_info = pack_info(ray_indices, n_rays)
    pk = packed_info.to(torch.int32).contiguous()
    if n_samples is not None:
        check_packed(pk, n_samples, what)
    return pk


def render_weight_from_alpha(alphas, *, ray_indices=None, packed_info=None, n_rays=None):
    """nerfacc 0.5.3 signature (models/volrend.py:851-855): flat alphas [S] -> (weights, trans)."""
    shape = alphas.shape
    w, t = _WeightFromAlpha.apply(_packed(ray_indices, packed_info, n_rays, alphas.numel(), "render_weight_from_alpha"),
                                  alphas.reshape(-1))
    return w.view(shape), t.view(shape)


def render_transmittance_from_alpha(alphas, *, ray_indices=None, packed_info=None, n_rays=None):
    shape = alphas.shape
    return _TransFromAlpha.apply(_packed(ray_indices, packed_info, n_rays, alphas.numel(), "render_transmittance_from_alpha"),
                                 alphas.reshape(-1)).view(shape)


@torch.no_grad()
def render_visibility(alphas, *, ray_indices=None, packed_info=None, n_rays=None,
                      early_stop_eps=1e-4, alpha_thre=0.0, zero_init=False):
    """lib/nerfacc/vol_rendering.py:452-520 -> bool [S].  ``zero_init``: entries that no ray's packed_info covers (the
    dummy tail of a capacity-sized buffer) read as False instead of being left unwritten."""
    pk = _packed(ray_indices, packed_info, n_rays, alphas.numel(), "render_visibility")
    a = _f32c(alphas.reshape(-1))
    require_device(pk, a)
    keep = (torch.zeros if zero_init else torch.empty)(a.numel(), dtype=torch.uint8, device=a.device)
    check(lib().rsdf_visibility_from_alpha(ptr(pk), ptr(a), pk.shape[0], float(early_stop_eps),
                                           float(alpha_thre), ptr(keep), stream_ptr()),
          "visibility_from_alpha")
    return keep.bool()


# ------------------------------------------------------------------------------------------------
# C2
# ------------------------------------------------------------------------------------------------
class _Accumulate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, packed_info, weights, values):
        w = _f32c(weights)
        v = None if values is None else _f32c(values)
        require_device(packed_info, w, v)
        n_rays = packed_info.shape[0]
        D = 1 if v is None else v.shape[-1]
        out = torch.empty(n_rays, D, dtype=torch.float32, device=w.device)
        check(lib().rsdf_accumulate_fwd(ptr(packed_info), ptr(w), ptr(v), n_rays, D, ptr(out),
                                        stream_ptr()), "accumulate_fwd")
        ctx.save_for_backward(packed_info, w, v)
        ctx.D = D
        return out

    @staticmethod
    def backward(ctx, go):
        packed_info, w, v = ctx.saved_tensors
        go = _f32c(go)
        need_w, need_v = ctx.needs_input_grad[1], ctx.needs_input_grad[2] and v is not None
        gw = torch.empty_like(w) if need_w else None
        gv = torch.empty_like(v) if need_v else None
        if need_w or need_v:
            check(lib().rsdf_accumulate_bwd(ptr(packed_info), ptr(w), ptr(v), ptr(go),
                                            packed_info.shape[0], ctx.D, ptr(gw), ptr(gv),
                                            stream_ptr()), "accumulate_bwd")
        return None, gw, gv


class _OpacityDepth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, packed_info, weights, t_starts, t_ends, want_mid):
        w, ts, te = _f32c(weights), _f32c(t_starts.reshape(-1)), _f32c(t_ends.reshape(-1))
        require_device(packed_info, w, ts, te)
        n_rays = packed_info.shape[0]
        opacity = torch.empty(n_rays, 1, dtype=torch.float32, device=w.device)
        depth = torch.empty(n_rays, 1, dtype=torch.float32, device=w.device)
        mid = torch.empty_like(ts) if want_mid else None
        check(lib().rsdf_opacity_depth_fwd(ptr(packed_info), ptr(w), ptr(ts), ptr(te), n_rays, ptr(opacity), ptr(depth),
                                           ptr(mid), stream_ptr()), "opacity_depth_fwd")
        ctx.save_for_backward(packed_info, ts, te)
        ctx.n_samples = w.shape[0]
        if want_mid:
            ctx.mark_non_differentiable(mid)
            return opacity, depth, mid
        return opacity, depth

    @staticmethod
    def backward(ctx, g_op, g_depth, *_):
        packed_info, ts, te = ctx.saved_tensors
        if not ctx.needs_input_grad[1] or (g_op is None and g_depth is None):
            return None, None, None, None, None
        g_op = None if g_op is None else _f32c(g_op)
        g_depth = None if g_depth is None else _f32c(g_depth)
        gw = torch.empty(ctx.n_samples, dtype=torch.float32, device=ts.device)   # (packed_info owns every sample)
        check(lib().rsdf_opacity_depth_bwd(ptr(packed_info), ptr(ts), ptr(te), ptr(g_op), ptr(g_depth),
                                           packed_info.shape[0], ptr(gw), stream_ptr()), "opacity_depth_bwd")
        return None, gw, None, None, None


def accumulate_opacity_depth(weights, t_starts, t_ends, *, ray_indices=None, packed_info=None, n_rays=None,
                             want_midpoints=False):
    """models/volrend.py:878-885 in one kernel each way: -> (opacity [n_rays,1], depth [n_rays,1]) =
    (accumulate_along_rays(weights, None), accumulate_along_rays(weights, (t_starts + t_ends)[..., None] / 2.0)), bit for
    bit, without the midpoint tensor -- or, with ``want_midpoints``, with it as a third (non-differentiable) output [S].
    t_starts / t_ends carry no gradient (they come from the marcher)."""
    pk = _packed(ray_indices, packed_info, n_rays, weights.numel(), "accumulate_opacity_depth")
    return _OpacityDepth.apply(pk, weights.reshape(-1), t_starts, t_ends, bool(want_midpoints))


def fold_normals():
    """The ray's normal map inside the opacity / depth pass (default) or through accumulate_along_rays (RSDF_FOLD_NORMALS=0)."""
    return os.environ.get("RSDF_FOLD_NORMALS", "1") != "0"


class _OpacityDepthNormal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, packed_info, weights, t_starts, t_ends, normals, want_mid):
        w, ts, te = _f32c(weights), _f32c(t_starts.reshape(-1)), _f32c(t_ends.reshape(-1))
        nm = _f32c(normals)
        require_device(packed_info, w, ts, te, nm)
        assert nm.shape == (w.shape[0], 3), "normals must be [n_samples, 3]"
        n_rays = packed_info.shape[0]
        dev = w.device
        opacity = torch.empty(n_rays, 1, dtype=torch.float32, device=dev)
        depth = torch.empty(n_rays, 1, dtype=torch.float32, device=dev)
        nmap = torch.empty(n_rays, 3, dtype=torch.float32, device=dev)
        mid = torch.empty_like(ts) if want_mid else None
        check(lib().rsdf_opacity_depth_normal_fwd(ptr(packed_info), ptr(w), ptr(ts), ptr(te), ptr(nm), n_rays, ptr(opacity),
                                                  ptr(depth), ptr(nmap), ptr(mid), stream_ptr()), "opacity_depth_normal_fwd")
        ctx.save_for_backward(packed_info, w, ts, te, nm)
        if want_mid:
            ctx.mark_non_differentiable(mid)
            return opacity, depth, nmap, mid
        return opacity, depth, nmap

    @staticmethod
    def backward(ctx, g_op, g_depth, g_n, *_):
        packed_info, w, ts, te, nm = ctx.saved_tensors
        need_w, need_n = ctx.needs_input_grad[1], ctx.needs_input_grad[4] and g_n is not None
        if (g_op is None and g_depth is None and g_n is None) or not (need_w or need_n):
            return None, None, None, None, None, None
        g_op = None if g_op is None else _f32c(g_op)
        g_depth = None if g_depth is None else _f32c(g_depth)
        g_n = None if g_n is None else _f32c(g_n)
        # zeros, not empty: capacity-mode callers hand in sample arrays whose tail no ray's packed_info covers
        gw = torch.zeros_like(w) if need_w else None
        gn = torch.zeros_like(nm) if need_n else None
        check(lib().rsdf_opacity_depth_normal_bwd(ptr(packed_info), ptr(w), ptr(ts), ptr(te), ptr(nm), ptr(g_op), ptr(g_depth),
                                                  ptr(g_n), packed_info.shape[0], ptr(gw), ptr(gn), stream_ptr()),
              "opacity_depth_normal_bwd")
        return None, gw, None, None, gn, None


def accumulate_opacity_depth_normal(weights, t_starts, t_ends, normals, *, ray_indices=None, packed_info=None, n_rays=None,
                                    want_midpoints=False):
    """models/volrend.py:875-885 in one kernel each way: -> (opacity [n_rays,1], depth [n_rays,1], normal map [n_rays,3]) =
    accumulate_opacity_depth(...) + (accumulate_along_rays(weights, normals),), bit for bit (with ``want_midpoints`` the
    [S] midpoints as a fourth, non-differentiable output)."""
    pk = _packed(ray_indices, packed_info, n_rays, weights.numel(), "accumulate_opacity_depth_normal")
    return _OpacityDepthNormal.apply(pk, weights.reshape(-1), t_starts, t_ends, normals, bool(want_midpoints))


def accumulate_along_rays(weights, values=None, *, ray_indices=None, packed_info=None, n_rays=None):
    """nerfacc 0.5.3 signature (models/volrend.py:871-885): weights [S], values [S,D] or None ->
    [n_rays, D or 1]."""
    pk = _packed(ray_indices, packed_info, n_rays, weights.numel(), "accumulate_along_rays")
    return _Accumulate.apply(pk, weights.reshape(-1), values)


# ------------------------------------------------------------------------------------------------
# H1 / H2
# ------------------------------------------------------------------------------------------------
class _HashGrid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, table, meta, n_active, include_xyz, xyz_scale, xyz_offset, fd7_eps_unit):
        xf, tb = _f32c(x), table.detach()
        require_device(xf, tb)
        assert tb.dtype == torch.float32 and tb.is_contiguous()
        n = xf.shape[0]
        LF = meta.n_levels * meta.n_features
        col = 3 if include_xyz else 0
        if fd7_eps_unit is not None and meta.n_features == 2 and n % 7 == 0 and n > 0:
            # x is the [S,7,3] stencil: gather with the stencil-aware kernel (8..32 merged corners per level
            # instead of 56, bit-identical values) and lay the level planes out as the reference-shaped rows
            S = n // 7
            Lv = meta.n_levels
            st = stream_ptr()
            # [S,7,3] interleaved -> tap-major, planes -> rows: layout kernels (as permuted torch copies these were 30 % of
            # the per-layer route's step)
            x7t = torch.empty(7, S, 3, dtype=torch.float32, device=xf.device)
            check(lib().rsdf_stencil_points_tap_major(ptr(xf), S, ptr(x7t), st), "stencil_points_tap_major")
            planes = L.workspace_f32("fd7.planes", (Lv, 7, S, 2), xf.device)     # (consumed by the rows kernel below)
            check(lib().rsdf_hashgrid_fwd_fd7(ptr(x7t), ptr(tb), ctypes.byref(meta), S, n_active, ptr(planes), st),
                  "hashgrid_fwd_fd7")
            out = torch.empty(n, col + LF, dtype=torch.float32, device=xf.device)
            check(lib().rsdf_stencil_planes_to_rows(ptr(planes), ptr(xf), S, Lv, n_active, ptr(out), col + LF, col,
                                                    int(include_xyz), float(xyz_scale), float(xyz_offset), st),
                  "stencil_planes_to_rows")
            ctx.x7t = x7t
        else:
            out = torch.empty(n, col + LF, dtype=torch.float32, device=xf.device)
            if _use_staged_gather(n):
                # large batches: level-major planes, then rows through LDS (bit-identical)
                nbytes = int(lib().rsdf_hashgrid_fwd_staged_scratch_bytes(ctypes.byref(meta), n, n_active))
                scratch = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=xf.device)
                check(lib().rsdf_hashgrid_fwd_staged(ptr(xf), ptr(tb), ctypes.byref(meta), n, n_active, ptr(out),
                                                     col + LF, col, int(include_xyz), float(xyz_scale),
                                                     float(xyz_offset), ptr(scratch), nbytes, stream_ptr()),
                      "hashgrid_fwd_staged")
            else:
                check(lib().rsdf_hashgrid_fwd(ptr(xf), ptr(tb), ctypes.byref(meta), n, n_active, ptr(out),
                                              col + LF, col, int(include_xyz), float(xyz_scale),
                                              float(xyz_offset), stream_ptr()), "hashgrid_fwd")
        ctx.save_for_backward(x if x.requires_grad else xf)
        ctx.meta, ctx.n_active, ctx.col, ctx.n_params = meta, n_active, col, tb.numel()
        ctx.fd7, ctx.xyz_scale = fd7_eps_unit, float(xyz_scale)
        ctx.table = table if x.requires_grad else None   # the input-gradient path re-reads the table
        return out

    @staticmethod
    def backward(ctx, gout):
        (x_in,) = ctx.saved_tensors
        xf, g = _f32c(x_in), _f32c(gout)
        dt = torch.zeros(ctx.n_params, dtype=torch.float32, device=xf.device)
        n = xf.shape[0]
        if ctx.fd7 is not None and ctx.meta.n_features == 2 and n % 7 == 0:
            # x holds [S,7,3] stencil taps: merge them in registers and reduce through LDS
            S = n // 7
            nbytes = int(lib().rsdf_hashgrid_bwd_fd7_scratch_bytes(ctypes.byref(ctx.meta), S,
                                                                   ctx.n_active, float(ctx.fd7)))
            if nbytes < 0:
                raise L.RiseSdfHipError("hashgrid_bwd_fd7: unsupported level layout")
            scratch = L.workspace("fd7.queues", nbytes, xf.device)
            # this entry point keeps the interleaved [S,7] row layout of the reference-shaped API;
            # re-lay out to the tap-major planes the stencil kernels consume (the fused field path
            # produces them directly)
            Lv = ctx.meta.n_levels
            x7t = getattr(ctx, "x7t", None)
            if x7t is None:
                x7t = torch.empty(7, S, 3, dtype=torch.float32, device=xf.device)
                check(lib().rsdf_stencil_points_tap_major(ptr(xf), S, ptr(x7t), stream_ptr()), "stencil_points_tap_major")
            dpl = L.workspace_f32("fd7.d_planes", (Lv, 7, S, 2), xf.device)
            check(lib().rsdf_stencil_rows_to_planes(ptr(g), g.shape[1], ctx.col, S, Lv, ptr(dpl), stream_ptr()),
                  "stencil_rows_to_planes")
            check(lib().rsdf_hashgrid_bwd_fd7(ptr(x7t), ptr(dpl), ctypes.byref(ctx.meta), S, ctx.n_active,
                                              float(ctx.fd7), ptr(dt), ptr(scratch), nbytes,
                                              stream_ptr()), "hashgrid_bwd_fd7")
        elif _use_binned(ctx.meta, n, ctx.n_active):
            # many plain points (the curvature term's 2.6e5 per training step): bins instead of per-corner atomics
            _scatter_binned(0, xf, g, g.shape[1], ctx.col, None, ctx.meta, ctx.n_active, dt)
        else:
            check(lib().rsdf_hashgrid_bwd(ptr(xf), ptr(g), ctypes.byref(ctx.meta), n, ctx.n_active,
                                          g.shape[1], ctx.col, ptr(dt), stream_ptr()), "hashgrid_bwd")
        dx = None
        if ctx.needs_input_grad[0]:
            # analytic normals / curvature: differentiable again (tcnn double backward)
            dx = _HashGridDx.apply(x_in, ctx.table, gout, ctx.meta, ctx.n_active, ctx.col)
            if ctx.col:
                dx = dx + gout[:, :3] * ctx.xyz_scale
        return dx, dt, None, None, None, None, None, None


BINNED_SCATTER_MIN_POINTS = 16384
STAGED_GATHER_MIN_POINTS = 1 << 15


def _use_staged_gather(n):
    """rsdf_hashgrid_fwd_staged from 2^15 points up (tools/bench_gather.py: ahead from 65536 points, the smallest size
    measured; below that both forms are launch latency); RSDF_GATHER=rows / staged forces either form for A/B."""
    mode = os.environ.get("RSDF_GATHER")
    if mode == "staged":
        return n > 0
    return mode != "rows" and n >= STAGED_GATHER_MIN_POINTS


def _scatter_binned(mode, xf, g, ld, col, gd, meta, n_active, dt):
    """dt += the table scatter of the generic backward (mode 0) / the input-gradient's backward (mode 1) through the
    bin-and-reduce queues (rsdf_hashgrid_scatter_binned) instead of per-corner float atomics."""
    n = xf.shape[0]
    nbytes = int(lib().rsdf_hashgrid_scatter_binned_scratch_bytes(ctypes.byref(meta), n, n_active))
    if nbytes < 0:
        raise L.RiseSdfHipError("hashgrid_scatter_binned: unsupported level layout")
    scratch = L.workspace("scatter.queues", nbytes, xf.device)       # (consumed inside this call)
    check(lib().rsdf_hashgrid_scatter_binned(mode, ptr(xf), ptr(g), ld, col, ptr(gd), ctypes.byref(meta), n, n_active,
                                             ptr(dt), ptr(scratch), nbytes, stream_ptr()), "hashgrid_scatter_binned")


def _use_binned(meta, n, n_active=None):
    """The bins serve F = 2 levels of at most 64 x 8192 = 2^19 entries (hashgrid_fd7.hip make_plan); a table with larger
    hashed levels (log2_hashmap_size 20..24 through the tcnn drop-in) keeps the per-corner atomic kernels, as does
    RSDF_SCATTER=atomics."""
    if meta.n_features != 2 or n < BINNED_SCATTER_MIN_POINTS or os.environ.get("RSDF_SCATTER") == "atomics":
        return False
    na = int(meta.n_levels) if n_active is None else int(n_active)
    return int(lib().rsdf_hashgrid_scatter_binned_scratch_bytes(ctypes.byref(meta), int(n), na)) >= 0


class _HashGridDx(torch.autograd.Function):
    """dx[S,3] = J_enc(x)^T dy over the active levels; backward = the hash grid's double backward."""

    @staticmethod
    def forward(ctx, x, table, dy, meta, n_active, col_off):
        xf, tb, g = _f32c(x), table.detach(), _f32c(dy)
        require_device(xf, tb, g)
        n = xf.shape[0]
        dx = torch.empty(n, 3, dtype=torch.float32, device=xf.device)
        check(lib().rsdf_hashgrid_dx(ptr(xf), ptr(tb), ctypes.byref(meta), n, n_active, ptr(g), g.shape[1],
                                     col_off, ptr(dx), stream_ptr()), "hashgrid_dx")
        ctx.save_for_backward(xf, tb, g)
        ctx.meta, ctx.n_active, ctx.col = meta, n_active, col_off
        return dx

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gdx):
        xf, tb, g = ctx.saved_tensors
        gd = _f32c(gdx)
        n = xf.shape[0]
        need_x, need_t, need_dy = ctx.needs_input_grad[:3]
        gx = torch.empty(n, 3, dtype=torch.float32, device=xf.device) if need_x else None
        dt = torch.zeros_like(tb) if need_t else None
        ddy = torch.zeros_like(g) if need_dy else None
        binned = need_t and _use_binned(ctx.meta, n, ctx.n_active)      # the table scatter through the bins, the rest as before
        if need_x or need_dy or (need_t and not binned):
            check(lib().rsdf_hashgrid_dx_bwd(ptr(xf), ptr(tb), ctypes.byref(ctx.meta), n, ctx.n_active, ptr(g),
                                             g.shape[1], ctx.col, ptr(gd), ptr(ddy), g.shape[1], ctx.col,
                                             None if binned else ptr(dt), ptr(gx), stream_ptr()), "hashgrid_dx_bwd")
        if binned:
            _scatter_binned(1, xf, g, g.shape[1], ctx.col, gd, ctx.meta, ctx.n_active, dt)
        return gx, dt, ddy, None, None, None


def hashgrid_dx(x, table, dy, meta, n_active_levels=None, col_off=0):
    """Input gradient of the hash-grid encoding: x [S,3] in [0,1], dy [S, col_off + L*F] -> [S,3] (unit-cube
    coordinates).  Differentiable w.r.t. x, table and dy."""
    na = meta.n_levels if n_active_levels is None else int(n_active_levels)
    return _HashGridDx.apply(x, table, dy, meta, na, col_off)


def hashgrid_encode(x, table, meta, n_active_levels=None, include_xyz=False, xyz_scale=2.0,
                    xyz_offset=-1.0, fd7_eps_unit=None):
    """x [S,3] in [0,1] -> [S, (3 +) L*F].  Differentiable w.r.t. ``table`` and (analytic-normal / curvature
    paths) ``x``.

    ``fd7_eps_unit`` (= eps / (2 radius)): promise that x is the [S/7, 7, 3] finite-difference
    stencil written by ``fd_points`` / ``fd_taps``; the backward then uses the stencil-merging,
    atomic-free scatter (rsdf_hashgrid_bwd_fd7)."""
    na = meta.n_levels if n_active_levels is None else int(n_active_levels)
    return _HashGrid.apply(x, table, meta, na, bool(include_xyz), xyz_scale, xyz_offset, fd7_eps_unit)


# ------------------------------------------------------------------------------------------------
# H3
# ------------------------------------------------------------------------------------------------
def _dw_db(N, K, has_bias, device):
    """One zero-fill for both accumulators of a layer (the training step at 4096 rays is launch-bound); db starts at a
    16-byte aligned offset so that a bias .grad is aligned like any other tensor."""
    off = (N * K + 3) // 4 * 4
    buf = torch.zeros(off + (N if has_bias else 0), dtype=torch.float32, device=device)
    return buf[:N * K].view(N, K), (buf[off:] if has_bias else None)


class _Linear(torch.autograd.Function):
    """y = act(x @ w^T + b) on the fp32 matrix cores; backward fused with the activation."""

    @staticmethod
    def forward(ctx, x, w, b, act, dx_cols, precision="fp32"):
        xf, wf = _f32c(x), _f32c(w)
        bf = None if b is None else _f32c(b)
        require_device(xf, wf, bf)
        n, K = xf.shape
        N = wf.shape[0]
        y = torch.empty(n, N, dtype=torch.float32, device=xf.device)
        check(L.mlp_fn("rsdf_linear_fwd", precision)(ptr(xf), K, ptr(wf), ptr(bf), n, K, N, act, ptr(y), N,
                                                     stream_ptr()), "linear_fwd")
        ctx.save_for_backward(xf, wf, y)
        ctx.act, ctx.has_bias, ctx.dx_cols, ctx.precision = act, b is not None, dx_cols, precision
        return y

    @staticmethod
    def backward(ctx, gy):
        xf, wf, y = ctx.saved_tensors
        gy = _f32c(gy)
        n, K = xf.shape
        N = wf.shape[0]
        st = stream_ptr()
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], \
            ctx.has_bias and ctx.needs_input_grad[2]
        dx = None
        k0, kout = (0, K) if ctx.dx_cols is None else ctx.dx_cols
        dx_win = None
        if need_x:
            # the kernel writes the [k0, k0+kout) column window in place (row stride K); columns
            # outside the window belong to non-differentiable inputs and are zero-filled
            dx = torch.empty(n, K, dtype=torch.float32, device=xf.device)
            if k0 > 0:
                dx[:, :k0].zero_()
            if k0 + kout < K:
                dx[:, k0 + kout:].zero_()
            dx_win = ctypes.c_void_p(dx.data_ptr() + 4 * k0)
        dw = db = None
        if need_w or need_b:
            dw, db = _dw_db(N, K, need_b, xf.device)
        fn = lambda name: L.mlp_fn(name, ctx.precision)   # noqa: E731
        if dw is not None and fn("rsdf_linear_bwd_fused_supported")(K, N) and os.environ.get("RSDF_LAYER_BWD") != "split":
            # 128-wide layers: one pass, dz never leaves the CU (mlp_layer_bwd.hip)
            check(fn("rsdf_linear_bwd_fused")(ptr(gy), ptr(y), N, ptr(xf), K, ptr(wf), n, K, N, ctx.act, k0, kout,
                                              dx_win, K, L.ACT_IDS["none"], ptr(dw), ptr(db), st), "linear_bwd_fused")
            return dx, dw, db, None, None, None
        # (no activation: dz IS dy -- nothing to write; an unaligned 49-column dz cost the drop-in route 0.1 s per step)
        dz = gy if ctx.act == L.ACT_IDS["none"] else torch.empty_like(gy)
        if dz is not gy or dx_win is not None:
            check(fn("rsdf_linear_bwd_input")(ptr(gy), ptr(y), N, ptr(wf), n, K, N, ctx.act, k0, kout,
                                              None if dz is gy else ptr(dz), dx_win, K, st), "linear_bwd_input")
        if dw is not None:
            check(fn("rsdf_linear_bwd_weight")(ptr(dz), N, ptr(xf), K, n, K, N, ptr(dw), ptr(db), st),
                  "linear_bwd_weight")
        return dx, dw, db, None, None, None


def linear(x, w, b=None, act="none", dx_cols=None, precision="fp32"):
    return _Linear.apply(x, w, b, L.ACT_IDS[act] if not isinstance(act, int) else act, dx_cols, precision)


class _MLPChain(torch.autograd.Function):
    """A whole VanillaMLP (models/network_utils.py:109-157: Linear -> act -> ... -> Linear [-> output act]) as ONE autograd
    node.  Forward: one rsdf_linear_fwd per layer.  Backward: the 128-wide layers run the one-pass kernel
    (rsdf_linear_bwd_fused), and when the layer below is Linear + ReLU the kernel writes THAT layer's dz = dx * (x > 0)
    straight away (x, the ReLU output, is on the CU already): the layer below then needs neither its activation pass nor
    its own output -- 1.5 instead of 2 KB per row and layer -- and autograd sees one node instead of one per layer."""

    @staticmethod
    def forward(ctx, x, dx_cols, acts, precision, *wb):
        xf = _f32c(x)
        ws = [_f32c(t) for t in wb[0::2]]
        bs = [None if t is None else _f32c(t) for t in wb[1::2]]
        require_device(xf, *ws)
        n = xf.shape[0]
        st = stream_ptr()
        h, ys = xf, []
        for w, b, act in zip(ws, bs, acts):
            N, K = w.shape
            assert h.shape[1] == K, "layer input width"
            y = torch.empty(n, N, dtype=torch.float32, device=xf.device)
            check(L.mlp_fn("rsdf_linear_fwd", precision)(ptr(h), K, ptr(w), ptr(b), n, K, N, act, ptr(y), N, st),
                  "linear_fwd")
            ys.append(y)
            h = y
        ctx.save_for_backward(xf, *ws, *ys)
        ctx.acts, ctx.dx_cols, ctx.n_layers, ctx.precision = tuple(acts), dx_cols, len(ws), precision
        ctx.has_bias = [b is not None for b in bs]
        return ys[-1]

    @staticmethod
    def backward(ctx, gy):
        nl = ctx.n_layers
        fn = lambda name: L.mlp_fn(name, ctx.precision)   # noqa: E731
        saved = ctx.saved_tensors
        xf, ws, ys = saved[0], saved[1:1 + nl], saved[1 + nl:]
        n = xf.shape[0]
        st = stream_ptr()
        relu, none = L.ACT_IDS["relu"], L.ACT_IDS["none"]
        split = os.environ.get("RSDF_LAYER_BWD") == "split"
        g, g_is_dz = _f32c(gy), False
        grads = [None] * (2 * nl)
        dx_in = None
        top, tail = nl - 1, None
        # frozen / evaluation weights (needs_input_grad says nobody wants dW or db of layer i): that layer runs the plain
        # input-gradient kernel and allocates, zero-fills and accumulates nothing
        need_w = [bool(ctx.needs_input_grad[4 + 2 * i] or (ctx.has_bias[i] and ctx.needs_input_grad[5 + 2 * i]))
                  for i in range(nl)]
        # ONE zero-fill for the weight / bias gradient accumulators of the whole network (the training step is launch-bound:
        # a zero-fill per layer was 28 launches per step); every accumulator starts at a 16-byte aligned offset
        offs, total = [], 0
        for i in range(nl):
            offs.append(total)
            if need_w[i]:
                N_, K_ = ws[i].shape
                total += (N_ * K_ + 3) // 4 * 4 + ((N_ + 3) // 4 * 4 if ctx.has_bias[i] else 0)
        pool = torch.zeros(total, dtype=torch.float32, device=xf.device) if total else None

        def layer_acc(i):
            N, K = ws[i].shape
            o, a = offs[i], (N * K + 3) // 4 * 4
            return pool[o:o + N * K].view(N, K), (pool[o + a:o + a + N] if ctx.has_bias[i] else None)

        # workspace of the one-pass layer backward (per-workgroup dW partials + a reduction instead of 4 M contended float
        # atomics per launch, include/risesdf_hip.h): one buffer for the whole chain -- its layers run one after the other on
        # this stream
        ws_bytes = 0
        if not split:
            for i in range(nl):
                if need_w[i]:
                    ws_bytes = max(ws_bytes, int(lib().rsdf_linear_bwd_fused_workspace_bytes(n, ws[i].shape[1], ws[i].shape[0])))
        wsp = torch.empty(ws_bytes // 4, dtype=torch.float32, device=xf.device) if ws_bytes else None

        # narrow output layer (<= 4 columns) on top of a 128-wide ReLU layer: only its dz and its weight gradient are
        # computed here; the layer below forms its own input gradient from that dz (rsdf_linear_bwd_fused_tail)
        if (not split and nl >= 2 and need_w[-1] and need_w[-2]
                and ws[-1].shape[0] <= 4 and ws[-1].shape[1] == 128 and ctx.acts[nl - 2] == relu
                and bool(fn("rsdf_linear_bwd_fused_supported")(ws[-2].shape[1], ws[-2].shape[0]))):
            w, y, xin = ws[-1], ys[-1], ys[-2]
            N, K = w.shape
            dzo = torch.empty_like(g)
            check(fn("rsdf_linear_bwd_input")(ptr(g), ptr(y), N, ptr(w), n, K, N, ctx.acts[-1], 0, K, ptr(dzo), None, K, st),
                  "linear_bwd_input")
            dw, db = layer_acc(nl - 1)
            check(fn("rsdf_linear_bwd_weight")(ptr(dzo), N, ptr(xin), K, n, K, N, ptr(dw), ptr(db), st), "linear_bwd_weight")
            grads[2 * top], grads[2 * top + 1] = dw, db
            tail, top = (dzo, N, w), nl - 2
        for i in range(top, -1, -1):
            w, y = ws[i], ys[i]
            xin = ys[i - 1] if i > 0 else xf
            N, K = w.shape
            act = none if g_is_dz else ctx.acts[i]
            yarg = None if g_is_dz else y
            need_dx = i > 0 or ctx.needs_input_grad[0]
            k0, kout = (0, K) if (i > 0 or ctx.dx_cols is None) else ctx.dx_cols
            dx = dx_win = None
            if need_dx:
                dx = torch.empty(n, K, dtype=torch.float32, device=xf.device)
                if k0 > 0:
                    dx[:, :k0].zero_()
                if k0 + kout < K:
                    dx[:, k0 + kout:].zero_()
                dx_win = ctypes.c_void_p(dx.data_ptr() + 4 * k0)
            if not need_w[i]:
                if need_dx:       # (dz is not needed by anyone here: not written)
                    check(fn("rsdf_linear_bwd_input")(ptr(g), ptr(yarg), N, ptr(w), n, K, N, act, k0, kout, None,
                                                      dx_win, K, st), "linear_bwd_input")
                g, g_is_dz = dx, False
                if i == 0:
                    dx_in = dx
                continue
            dw, db = layer_acc(i)
            fused = not split and bool(fn("rsdf_linear_bwd_fused_supported")(K, N))
            prev_relu = fused and i > 0 and ctx.acts[i - 1] == relu
            if tail is not None:
                dzo, n2, w2 = tail
                tail = None
                if dx is None:   # (a one-hidden-layer network whose input needs no gradient: still wants a dx buffer)
                    dx = torch.empty(n, K, dtype=torch.float32, device=xf.device)
                    dx_win, k0, kout = ptr(dx), 0, K
                check(fn("rsdf_linear_bwd_fused_tail_ws")(ptr(dzo), n2, ptr(w2), ptr(y), N, ptr(xin), K, ptr(w), n, K, N,
                                                          ctx.acts[i], k0, kout, dx_win, K, relu if prev_relu else none,
                                                          ptr(dw), ptr(db), ptr(wsp), ws_bytes, st), "linear_bwd_fused_tail")
                if not need_dx:
                    dx = None
            elif fused:
                check(fn("rsdf_linear_bwd_fused_ws")(ptr(g), ptr(yarg), N, ptr(xin), K, ptr(w), n, K, N, act, k0, kout,
                                                     dx_win, K, relu if prev_relu else none, ptr(dw), ptr(db), ptr(wsp),
                                                     ws_bytes, st), "linear_bwd_fused")
            else:
                dz = g if act == none else torch.empty_like(g)          # (no activation: dz is dy, nothing to write)
                if dz is not g or dx_win is not None:
                    check(fn("rsdf_linear_bwd_input")(ptr(g), ptr(yarg), N, ptr(w), n, K, N, act, k0, kout,
                                                      None if dz is g else ptr(dz), dx_win, K, st), "linear_bwd_input")
                check(fn("rsdf_linear_bwd_weight")(ptr(dz), N, ptr(xin), K, n, K, N, ptr(dw), ptr(db), st),
                      "linear_bwd_weight")
            grads[2 * i], grads[2 * i + 1] = dw, db
            g, g_is_dz = dx, prev_relu
            if i == 0:
                dx_in = dx
        return (dx_in, None, None, None, *grads)


_PAIR_PACK_CACHE = {}      # "last": (key, weakref to the input rows, image): see _MLPPairChain.forward


class _MLPPairChain(torch.autograd.Function):
    """A radiance network (models/texture.py:237-327: Linear(K,128) ReLU [Linear(128,128) ReLU]x1|3 Linear(128,N2) [act]) on the
    layer-PAIR kernels (csrc/mlp_pair.hip): two hidden layers per kernel in the x2 number format, the odd activation never
    written (forward) and recomputed (backward), a pair's input / output through HBM once each way as a pair image.  The narrow
    output layer stays on the per-layer kernels, on fp32 rows of the last hidden activation.  Per row of a 4-hidden-layer
    network: ~2.5 KB forward + ~4 KB backward against 3.9 + 8 with one kernel per layer."""

    @staticmethod
    def forward(ctx, x, x2, dx_cols, acts, parts, *wb):
        """``x2`` (nullable): the network's input is cat([x, x2], -1) (models/texture.py:299-313: [feature, encoding]) and is
        never materialised -- rsdf_pair_pack2 writes the pair image from the two sources.  ``parts``: 2 = fp32-equivalent (two
        fp16 parts, three products), 1 = the network's 16-bit mode (``precision: fp16 / bf16``: operands rounded once to fp16,
        one product; rsdf_pair_fwd16 / _bwd16)."""
        xf = _f32c(x)
        xf2 = None if x2 is None else _f32c(x2)
        ws = [_f32c(t) for t in wb[0::2]]
        bs = [_f32c(t) for t in wb[1::2]]
        require_device(xf, xf2, *ws, *bs)
        n, K1 = xf.shape
        K2 = 0 if xf2 is None else xf2.shape[1]
        K = K1 + K2
        dev, st = xf.device, stream_ptr()
        nh = len(ws) - 1                                  # hidden layers: 2 or 4
        stt = ptr(L.status(dev))
        img_bytes = int(lib().rsdf_pair_image_bytes(n))
        # the albedo / roughness / metallic networks of models/texture.py:303-324 read the SAME input tensor: one image serves them
        # (inside an autograd Function every call sees a fresh alias of the caller's tensor: the key is the storage address, the
        # shared version counter and the shape; the entry pins the rows it was packed from, so the address cannot be reused
        # while it is cached, and the next pack replaces it)
        key = (xf.data_ptr(), xf._version, None if xf2 is None else (xf2.data_ptr(), xf2._version), n, K1, K2, dev.index,
               int(st.value or 0))
        hit = _PAIR_PACK_CACHE.get("last")
        if hit is not None and hit[0] == key:
            imgs = [hit[2]]
        else:
            imgs = [torch.empty(img_bytes, dtype=torch.uint8, device=dev)]
            check(lib().rsdf_pair_pack2(ptr(xf), K1, K1, ptr(xf2), K2, K2, n, ptr(imgs[0]), stt, st), "pair_pack")
            _PAIR_PACK_CACHE["last"] = (key, (xf, xf2), imgs[0])
        h_last = torch.empty(n, 128, dtype=torch.float32, device=dev)
        N2 = ws[-1].shape[0]
        y = torch.empty(n, N2, dtype=torch.float32, device=dev)
        # the narrow output layer rides in the last pair's kernel (N2 <= 8, no activation or the sigmoid the texture networks
        # fuse into it): no pass over the last hidden activation for it, forward (y) or backward (d h_last)
        fold = N2 <= 8 and acts[-1] in (L.ACT_IDS["none"], L.ACT_IDS["sigmoid"])
        # (forward: two waves form y = W_out hb from hb's LDS image with 12 matrix instructions per tile -- a first version on the
        # vector ALU with a cross-wave sum cost the memory-bound kernel more (+130 ms of a c2 step) than the pass it replaced
        # (113 ms).  RSDF_PAIR_FOLD_FWD=0: the per-layer kernel, for A/B and tests.)
        fold_fwd = fold and os.environ.get("RSDF_PAIR_FOLD_FWD", "1") != "0"
        for p in range(nh // 2):
            last = p == nh // 2 - 1
            out_img = None if last else torch.empty(img_bytes, dtype=torch.uint8, device=dev)
            lf = last and fold_fwd
            pair_fwd = lib().rsdf_pair_fwd16 if parts == 1 else lib().rsdf_pair_fwd
            check(pair_fwd(ptr(imgs[p]), K if p == 0 else 128, ptr(ws[2 * p]), ptr(bs[2 * p]), ptr(ws[2 * p + 1]),
                           ptr(bs[2 * p + 1]), n, ptr(out_img), ptr(h_last) if last else None,
                           ptr(ws[-1]) if lf else None, ptr(bs[-1]) if lf else None, N2 if lf else 0,
                           acts[-1] if lf else 0, ptr(y) if lf else None, stt, st), "pair_fwd")
            if not last:
                imgs.append(out_img)
        if not fold_fwd:
            check(lib().rsdf_linear_fwd(ptr(h_last), 128, ptr(ws[-1]), ptr(bs[-1]), n, 128, N2, acts[-1], ptr(y), N2, st),
                  "linear_fwd")
        ctx.save_for_backward(*imgs, h_last, y, *ws, *bs)
        ctx.n_imgs, ctx.nh, ctx.K, ctx.acts, ctx.dx_cols, ctx.fold = len(imgs), nh, K, tuple(acts), dx_cols, fold
        ctx.K1, ctx.parts = K1, parts
        return y

    @staticmethod
    def backward(ctx, gy):
        sv = ctx.saved_tensors
        ni, nh, K = ctx.n_imgs, ctx.nh, ctx.K
        pair_bwd = lib().rsdf_pair_bwd16 if ctx.parts == 1 else lib().rsdf_pair_bwd
        imgs, h_last, y = sv[:ni], sv[ni], sv[ni + 1]
        ws, bs = sv[ni + 2:ni + 2 + nh + 1], sv[ni + 3 + nh:]
        n = h_last.shape[0]
        dev, st = h_last.device, stream_ptr()
        g = _f32c(gy)
        N2 = ws[-1].shape[0]
        # ONE zero-fill for every weight / bias gradient accumulator of the network (the kernels accumulate)
        sizes = []
        for w in ws:
            sizes += [(w.numel() + 3) // 4 * 4, (w.shape[0] + 3) // 4 * 4]
        pool = torch.zeros(sum(sizes) + 4, dtype=torch.float32, device=dev)
        grads, o = [], 0
        for w, sw, sb in zip(ws, sizes[0::2], sizes[1::2]):
            grads += [pool[o:o + w.numel()].view_as(w), pool[o + sw:o + sw + w.shape[0]]]
            o += sw + sb
        bounds = pool[o:o + 4].view(torch.int32)          # [0..1]: the top pair's bound + scratch, [2]: the lower pair's
        # ---- narrow output layer on the per-layer kernels: dz_out, d h_last [n,128] rows, dW_out, db_out
        dzo = g if ctx.acts[-1] == L.ACT_IDS["none"] else torch.empty_like(g)
        fold = ctx.fold
        dh = None if fold else torch.empty(n, 128, dtype=torch.float32, device=dev)
        if dzo is not g or dh is not None:       # (folded: only dz_out = g act'(y), [n, N2]; d h_last is formed inside the pair kernel)
            check(lib().rsdf_linear_bwd_input(ptr(g), ptr(y), N2, ptr(ws[-1]), n, 128, N2, ctx.acts[-1], 0, 128,
                                              None if dzo is g else ptr(dzo), ptr(dh), 128, st), "linear_bwd_input")
        if not fold:
            check(lib().rsdf_linear_bwd_weight(ptr(dzo), N2, ptr(h_last), 128, n, 128, N2, ptr(grads[-2]), ptr(grads[-1]), st),
                  "linear_bwd_weight")
        # (folded: dW_out rides in the top pair's kernel, which reads h_last for the ReLU mask anyway, and db_out = the column
        #  sums of dz_out come out of the pass that finds its maximum)
        check(lib().rsdf_pair_bound_from_out_layer(ptr(dzo), n, N2, ptr(ws[-1]), ptr(bounds), ptr(grads[-1]) if fold else None,
                                                   st), "pair_bound")
        # ---- the pairs, top down
        need1, need2 = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_dx = need1 or need2
        k0, kout = (0, K) if ctx.dx_cols is None else ctx.dx_cols
        if not need2:                            # (only the first source's columns: fewer dx slabs are computed at all)
            kout = min(k0 + kout, ctx.K1) - k0
        dx_in = dx_second = None
        gcur, masked, bound = dh, 0, bounds
        for p in range(nh // 2 - 1, -1, -1):
            Kp = K if p == 0 else 128
            if p > 0:
                dx = torch.empty(n, 128, dtype=torch.float32, device=dev)
                win, ld, ko, relu, amax = ptr(dx), 128, 128, 1, ctypes.c_void_p(bounds.data_ptr() + 8)
            elif need_dx:
                # only the first source wants a gradient (the radiance networks' [feature | encoding] input: the encodings carry
                # none): its columns as a CONTIGUOUS [n, K1] tensor -- autograd sums the three material networks' gradients of
                # the shared feature tensor, and on a [:, :K1] view of [n, K] rows that sum (and the zero-fill of the unused
                # columns) ran on torch's strided elementwise kernels: 190 ms of a 4.5 s config[2] step
                Kd = K if need2 else ctx.K1
                # both sources want one (the specular network's [feature | SH(reflected direction)]): two contiguous tensors
                # from the kernel (dx2), for the same reason -- the feature part joins autograd's sum, the other part goes to
                # the encoding's backward, which would copy a strided view first
                split = (need1 and need2 and ctx.dx_cols is None and 0 < ctx.K1 < K and ctx.K1 % 4 == 0
                         and os.environ.get("RSDF_PAIR_SPLIT_DX", "1") != "0")
                dx = torch.empty(n, ctx.K1 if split else Kd, dtype=torch.float32, device=dev)
                dx_second = torch.empty(n, K - ctx.K1, dtype=torch.float32, device=dev) if split else None
                if not split and k0 + kout < Kd:
                    dx[:, k0 + kout:].zero_()
                win, ld, ko, relu, amax = ptr(dx), dx.shape[1], k0 + kout, 0, None
            else:
                dx, win, ld, ko, relu, amax = None, None, 0, 0, 0, None
            second = dx_second if (p == 0 and need_dx) else None
            # (the top pair takes its ReLU mask from the forward's own h_last rows: no hb recompute, the lean kernel variant)
            top_fold = masked == 0 and fold
            check(pair_bwd(ptr(imgs[p]), Kp, ptr(ws[2 * p]), ptr(bs[2 * p]), ptr(ws[2 * p + 1]), ptr(bs[2 * p + 1]), n,
                                      None if top_fold else ptr(gcur), masked, ptr(h_last) if masked == 0 else None,
                                      ptr(dzo) if top_fold else None, ptr(ws[-1]) if top_fold else None, N2 if top_fold else 0,
                                      ptr(grads[-2]) if top_fold else None,
                                      ptr(bound) if masked == 0 else ctypes.c_void_p(bounds.data_ptr() + 8),
                                      win, ld, ko, ptr(second), 0 if second is None else second.shape[1],
                                      0 if second is None else ctx.K1, relu, amax, ptr(grads[4 * p]), ptr(grads[4 * p + 1]),
                                      ptr(grads[4 * p + 2]), ptr(grads[4 * p + 3]), st), "pair_bwd")
            if p == 0:
                if dx is not None and k0 > 0:
                    dx[:, :k0].zero_()
                dx_in = dx
            gcur, masked = dx, 1
        K1 = ctx.K1
        if dx_in is not None and need_dx and dx_second is not None:
            return (dx_in, dx_second, None, None, None, *grads)
        d1 = dx_in[:, :K1] if (dx_in is not None and need1) else None
        d2 = dx_in[:, K1:] if (dx_in is not None and need2 and K1 < K) else None
        if d1 is not None and dx_in.shape[1] == K1:
            d1 = dx_in
        return (d1, d2, None, None, None, *grads)


def pair_chain_ok(x, ws, bs, acts, precision, x2=None):
    """The radiance networks of models/texture.py:237-327 as the reference builds them: 2 or 4 hidden layers of 128 with ReLU
    and biases, at most 128 inputs; fp32 (two fp16 parts) or the network's 16-bit mode (``precision: bf16 / fp16``: one fp16
    part, round 6).  ``RSDF_PAIR=0`` keeps one kernel per layer, ``RSDF_PAIR16=0`` only for the 16-bit mode."""
    relu = L.ACT_IDS["relu"]
    nh = len(ws) - 1
    K = x.shape[1] + (0 if x2 is None else x2.shape[1]) if x.dim() == 2 else -1
    if precision in ("bf16", "fp16") and os.environ.get("RSDF_PAIR16", "1") == "0":      # A/B: the per-layer _bf16 kernels
        return False
    return (precision in (None, "fp32", "bf16", "fp16") and os.environ.get("RSDF_PAIR", "1") != "0" and not L.range_free("pair")
            and nh in (2, 4)
            and os.environ.get("RSDF_LAYER_BWD") != "split"
            and x.dim() == 2 and x.shape[0] > 0 and 1 <= K <= 128 and ws[0].shape == (128, K)
            and (x2 is None or (x2.dim() == 2 and x2.shape[0] == x.shape[0] and x2.is_cuda))
            and all(tuple(w.shape) == (128, 128) for w in ws[1:nh]) and ws[nh].shape[1] == 128
            and all(b is not None for b in bs) and all(a == relu for a in acts[:nh]) and x.is_cuda)


def mlp_chain(x, layers, acts, dx_cols=None, precision="fp32", x2=None):
    """``layers`` = [(W [out,in], b [out] or None)], ``acts`` = activation name per layer; see _MLPChain.
    ``precision``: 'fp32' (fp32-equivalent split products) or 'bf16' (one bf16 product, fp32 accumulate; opt-in).
    ``x2``: the network's input is cat([x, x2], -1); the pair kernels never materialise it, every other route concatenates."""
    flat = []
    for w, b in layers:
        flat += [w, b]
    act_ids = tuple(L.ACT_IDS[a] if not isinstance(a, int) else a for a in acts)
    if pair_chain_ok(x, [w for w, _ in layers], [b for _, b in layers], act_ids, precision, x2):
        return _MLPPairChain.apply(x.float(), None if x2 is None else x2.float(), dx_cols, act_ids,
                                   1 if precision in ("bf16", "fp16") else 2, *flat)
    if x2 is not None:
        x = torch.cat([x, x2.to(x.dtype)], dim=-1)
    return _MLPChain.apply(x, dx_cols, act_ids, precision, *flat)


class _WeightNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, v):
        gf, vf = _f32c(g).reshape(-1), _f32c(v)
        require_device(gf, vf)
        w = torch.empty_like(vf)
        check(lib().rsdf_weight_norm_fwd(ptr(gf), ptr(vf), vf.shape[0], vf.shape[1], ptr(w),
                                         stream_ptr()), "weight_norm_fwd")
        ctx.save_for_backward(gf, vf)
        ctx.gshape = g.shape
        return w

    @staticmethod
    def backward(ctx, dw):
        gf, vf = ctx.saved_tensors
        dw = _f32c(dw)
        dg, dv = torch.empty_like(gf), torch.empty_like(vf)
        check(lib().rsdf_weight_norm_bwd(ptr(gf), ptr(vf), ptr(dw), vf.shape[0], vf.shape[1], ptr(dg),
                                         ptr(dv), stream_ptr()), "weight_norm_bwd")
        return dg.view(ctx.gshape), dv


def weight_norm(g, v):
    """torch.nn.utils.weight_norm(dim=0): w = v * (g / ||v||_row)."""
    return _WeightNorm.apply(g, v)


# ------------------------------------------------------------------------------------------------
# P1 / H4 / A1
# ------------------------------------------------------------------------------------------------
@torch.no_grad()
def fd_points(rays_o, rays_d, ray_indices, t_starts, t_ends, radius, eps, want_positions=False,
              tap_major=False, want_taps=True):
    """Sample midpoints + six clamped FD taps, AABB-contracted to [0,1]: x_unit [S,7,3], or the
    tap-major [7,S,3] the fused stencil kernels consume.  ``want_taps=False`` (with ``want_positions``): -> (None, positions) --
    the x2 stencil kernels derive the taps from the positions, so the 84 B per sample of x_unit are neither written nor kept
    for the backward."""
    o, d, ts, te = _f32c(rays_o), _f32c(rays_d), _f32c(t_starts), _f32c(t_ends)
    ri = ray_indices.contiguous()
    require_device(o, d, ri, ts, te)
    n = ri.numel()
    assert want_taps or want_positions
    shape = (7, n, 3) if tap_major else (n, 7, 3)
    xu = torch.empty(shape, dtype=torch.float32, device=o.device) if want_taps else None
    pos = torch.empty(n, 3, dtype=torch.float32, device=o.device) if want_positions else None
    check(lib().rsdf_fd_points(ptr(o), ptr(d), ptr(ri), ptr(ts), ptr(te), n, float(radius), float(eps),
                               ptr(xu), ptr(pos), int(tap_major), stream_ptr()), "fd_points")
    return (xu, pos) if want_positions else xu


@torch.no_grad()
def fd_taps(points, radius, eps):
    """World-space points [S,3] -> centre + six clamped FD taps, contracted: x_unit [S,7,3]."""
    p = _f32c(points)
    require_device(p)
    xu = torch.empty(p.shape[0], 7, 3, dtype=torch.float32, device=p.device)
    check(lib().rsdf_fd_taps(ptr(p), p.shape[0], float(radius), float(eps), ptr(xu), stream_ptr()),
          "fd_taps")
    return xu


class _FDGradient(torch.autograd.Function):
    """out7 [7S, ld] -> sdf [S] (= column 0 of the centre tap), grad [S,3]."""

    @staticmethod
    def forward(ctx, out7, eps):
        o7 = out7.detach()
        require_device(o7)
        assert o7.dtype == torch.float32 and o7.is_contiguous() and o7.dim() == 2
        n = o7.shape[0] // 7
        sdf = torch.empty(n, dtype=torch.float32, device=o7.device)
        grad = torch.empty(n, 3, dtype=torch.float32, device=o7.device)
        check(lib().rsdf_fd_gradient_fwd(ptr(o7), o7.shape[1], float(eps), n, ptr(sdf), ptr(grad),
                                         stream_ptr()), "fd_gradient_fwd")
        ctx.eps, ctx.shape = float(eps), o7.shape
        ctx.set_materialize_grads(False)
        return sdf, grad

    @staticmethod
    def backward(ctx, g_sdf, g_grad):
        if g_sdf is None and g_grad is None:
            return None, None
        d = torch.zeros(ctx.shape, dtype=torch.float32, device=(g_sdf if g_sdf is not None else g_grad).device)
        gs = None if g_sdf is None else _f32c(g_sdf)
        gg = None if g_grad is None else _f32c(g_grad)
        check(lib().rsdf_fd_gradient_bwd(ptr(gs), ptr(gg), ctx.eps, ctx.shape[0] // 7, ptr(d),
                                         ctx.shape[1], stream_ptr()), "fd_gradient_bwd")
        return d, None


def fd_gradient(out7, eps):
    return _FDGradient.apply(out7, eps)


class _NeusAlphaFD(torch.autograd.Function):
    """sdf7 [7S, ld] (column 0 = SDF of the 7 taps) -> sdf [S], grad [S,3], normal [S,3], alpha [S]."""

    @staticmethod
    def forward(ctx, out7, variance, rays_d, ray_indices, t_starts, t_ends, cos_anneal_ratio, eps,
                tap_major):
        o7, var = out7.detach(), _f32c(variance).reshape(1)
        d, ri, ts, te = _f32c(rays_d), ray_indices.contiguous(), _f32c(t_starts), _f32c(t_ends)
        require_device(o7, var, d, ri, ts, te)
        assert o7.dtype == torch.float32 and o7.is_contiguous() and o7.dim() == 2
        n = ri.numel()
        if tap_major:
            assert tuple(o7.shape) == (7, n), "tap-major SDF stencil must be [7, S]"
            ld = -1  # RSDF_TAP_MAJOR
        else:
            assert o7.shape[0] == 7 * n
            ld = o7.shape[1]
        ctx.ld = ld
        dev = o7.device
        sdf = torch.empty(n, dtype=torch.float32, device=dev)
        grad = torch.empty(n, 3, dtype=torch.float32, device=dev)
        normal = torch.empty(n, 3, dtype=torch.float32, device=dev)
        alpha = torch.empty(n, dtype=torch.float32, device=dev)
        check(lib().rsdf_neus_alpha_fd_fwd(ptr(o7), ld, ptr(d), ptr(ri), ptr(ts), ptr(te),
                                           ptr(var), float(cos_anneal_ratio), float(eps), n,
                                           ptr(sdf), ptr(grad), ptr(normal), ptr(alpha), stream_ptr()),
              "neus_alpha_fd_fwd")
        ctx.save_for_backward(o7, var, d, ri, ts, te)
        ctx.car, ctx.eps, ctx.vshape = float(cos_anneal_ratio), float(eps), variance.shape
        ctx.set_materialize_grads(False)   # unused outputs arrive as None instead of [S]-sized zero fills
        return sdf, grad, normal, alpha

    @staticmethod
    def backward(ctx, g_sdf, g_grad, g_normal, g_alpha):
        if g_sdf is None and g_grad is None and g_normal is None and g_alpha is None:
            return (None,) * 9
        o7, var, d, ri, ts, te = ctx.saved_tensors
        n = ri.numel()
        ld = ctx.ld
        # interleaved: only column 0 receives gradient; tap-major: every element is written
        d_out7 = torch.empty_like(o7) if ld < 0 else torch.zeros_like(o7)
        d_var = torch.zeros(1, dtype=torch.float32, device=o7.device)
        cg = lambda t: None if t is None else _f32c(t)
        gs, gg, gn, ga = cg(g_sdf), cg(g_grad), cg(g_normal), cg(g_alpha)
        check(lib().rsdf_neus_alpha_fd_bwd(ptr(o7), ld, ptr(d), ptr(ri), ptr(ts), ptr(te), ptr(var),
                                           ctx.car, ctx.eps, n, ptr(ga), ptr(gn), ptr(gs), ptr(gg),
                                           ptr(d_out7), ld, ptr(d_var), stream_ptr()),
              "neus_alpha_fd_bwd")
        return d_out7, d_var.view(ctx.vshape), None, None, None, None, None, None, None


def neus_alpha_fd(out7, variance, rays_d, ray_indices, t_starts, t_ends, cos_anneal_ratio, eps,
                  tap_major=False):
    """out7: [7S, ld] rows 7i+t (column 0 = SDF), or with ``tap_major`` the [7, S] SDF stencil."""
    return _NeusAlphaFD.apply(out7, variance, rays_d, ray_indices, t_starts, t_ends,
                              cos_anneal_ratio, eps, bool(tap_major))


class _NeusAlpha(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sdf, normal, dirs, dists, variance, cos_anneal_ratio):
        s, nm, dr, ds = _f32c(sdf).reshape(-1), _f32c(normal), _f32c(dirs), _f32c(dists).reshape(-1)
        var = _f32c(variance).reshape(1)
        require_device(s, nm, dr, ds, var)
        alpha = torch.empty_like(s)
        check(lib().rsdf_neus_alpha_fwd(ptr(s), ptr(nm), ptr(dr), ptr(ds), ptr(var),
                                        float(cos_anneal_ratio), s.numel(), ptr(alpha), stream_ptr()),
              "neus_alpha_fwd")
        ctx.save_for_backward(s, nm, dr, ds, var)
        ctx.car, ctx.vshape, ctx.sshape = float(cos_anneal_ratio), variance.shape, sdf.shape
        return alpha

    @staticmethod
    def backward(ctx, ga):
        s, nm, dr, ds, var = ctx.saved_tensors
        ga = _f32c(ga)
        d_sdf, d_n = torch.empty_like(s), torch.empty_like(nm)
        d_var = torch.zeros(1, dtype=torch.float32, device=s.device)
        check(lib().rsdf_neus_alpha_bwd(ptr(s), ptr(nm), ptr(dr), ptr(ds), ptr(var), ctx.car,
                                        s.numel(), ptr(ga), ptr(d_sdf), ptr(d_n), ptr(d_var),
                                        stream_ptr()), "neus_alpha_bwd")
        return d_sdf.view(ctx.sshape), d_n, None, None, d_var.view(ctx.vshape), None


def neus_alpha(sdf, normal, dirs, dists, variance, cos_anneal_ratio=1.0):
    """get_alpha (models/split_mixed_occ.py:151-177) with the VarianceNetwork folded in."""
    return _NeusAlpha.apply(sdf, normal, dirs, dists, variance, cos_anneal_ratio)


# ------------------------------------------------------------------------------------------------
# M2 / N2
# ------------------------------------------------------------------------------------------------
@torch.no_grad()
def occ_alpha(sdf, variance, render_step_size):
    """occ_eval_fn (models/split_mixed_occ.py:108-119, models/neus.py:101-111): [n] SDF -> [n,1] alpha with
    cos == -1 and dists == render_step_size; inv_s = clip(exp(10 variance), 1e-6, 1e6) on the device."""
    s = _f32c(sdf.reshape(-1))
    v = _f32c(variance.reshape(-1))
    require_device(s, v)
    out = torch.empty_like(s)
    check(lib().rsdf_neus_occ_alpha(ptr(s), ptr(v), float(render_step_size), s.numel(), ptr(out), stream_ptr()),
          "neus_occ_alpha")
    return out.view(-1, 1)


@torch.no_grad()
def occ_cell_points(indices, jitter, roi, resolution):
    """lib/nerfacc/grid.py:213-222: world-space sample point of each listed cell (all cells when ``indices`` is
    None): (coords + jitter) / res * (roi_max - roi_min) + roi_min."""
    j, r = _f32c(jitter), _f32c(roi)
    idx = None if indices is None else indices.to(torch.int64).contiguous()
    require_device(j, r, idx)
    n = j.shape[0]
    x = torch.empty(n, 3, dtype=torch.float32, device=j.device)
    rx, ry, rz = (int(v) for v in resolution)
    check(lib().rsdf_occ_cell_points(ptr(idx), ptr(j), ptr(r), rx, ry, rz, n, ptr(x), stream_ptr()), "occ_cell_points")
    return x


@torch.no_grad()
def occ_update(occs, binary_u8, indices, occ, ema_decay, occ_thre):
    """lib/nerfacc/grid.py:229-238 in place: occs[idx] = max(occs[idx] * decay, occ); binary = occs > min(mean, thre)."""
    o = _f32c(occ).reshape(-1)
    idx = None if indices is None else indices.to(torch.int64).contiguous()
    require_device(occs, binary_u8, o, idx)
    assert occs.dtype == torch.float32 and occs.is_contiguous() and binary_u8.dtype == torch.uint8
    n_cells = occs.numel()
    scratch = torch.empty(int(lib().rsdf_occ_update_scratch_bytes(n_cells)), dtype=torch.uint8, device=occs.device)
    check(lib().rsdf_occ_update(ptr(idx), ptr(o), o.numel(), float(ema_decay), float(occ_thre), n_cells, ptr(occs),
                                ptr(binary_u8), ptr(scratch), stream_ptr()), "occ_update")


@torch.no_grad()
def gen_rays(index, y, x, directions, c2w, images=None, fg_masks=None, background_color=None, apply_mask=False):
    """systems/split_occ.py:66-81,103,113-116: -> (rays [n,6], rgb [n,C] or None, fg_mask [n] or None)."""
    idx, yy, xx = (t.to(torch.int64).contiguous() for t in (index, y, x))
    d, m = _f32c(directions), _f32c(c2w)
    img = None if images is None else _f32c(images)
    msk = None if fg_masks is None else _f32c(fg_masks)
    bg = None if background_color is None else _f32c(background_color)
    require_device(idx, yy, xx, d, m, img, msk, bg)
    n = yy.numel()
    H, W = d.shape[-3], d.shape[-2]
    rays = torch.empty(n, 6, dtype=torch.float32, device=d.device)
    C = 0 if img is None else img.shape[-1]
    rgb = None if img is None else torch.empty(n, C, dtype=torch.float32, device=d.device)
    fg = None if msk is None else torch.empty(n, dtype=torch.float32, device=d.device)
    check(lib().rsdf_gen_rays(ptr(idx), idx.numel(), ptr(yy), ptr(xx), ptr(d), int(d.dim() == 4), ptr(m), ptr(img), C,
                              ptr(msk), ptr(bg), int(bool(apply_mask)), H, W, n, ptr(rays), ptr(rgb), ptr(fg),
                              stream_ptr()), "gen_rays")
    return rays, rgb, fg
