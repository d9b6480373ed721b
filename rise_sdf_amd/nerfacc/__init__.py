"""Drop-in for the nerfacc surface RISE-SDF uses, backed by librisesdf_hip.so.

Two API generations are mirrored, because the reference uses both:

* nerfacc 0.5.3 (pip, ``requirements.txt:4``) -- the names the live models import
  (``models/split_mixed_occ.py:12-15``, ``models/neus.py:11-17``, ``models/volrend.py:10-14``):
  ``OccGridEstimator``, ``render_weight_from_alpha``, ``accumulate_along_rays``,
  ``ray_aabb_intersect``.  Its traversal kernel is not vendored, so the sample placement follows
  the vendored 0.3.5 marcher (the only one with source) behind the 0.5.3 signatures.
* vendored nerfacc 0.3.5 (``lib/nerfacc``): ``ContractionType``, ``OccupancyGrid``,
  ``ray_marching``, ``pack_info``, ``unpack_info``, ``render_visibility``,
  ``render_transmittance_from_alpha``.

Random numbers: the reference draws stratified jitter and cell jitter on device
(``lib/nerfacc/ray_marching.py:157-158``, ``grid.py:184-186,215-217``).  Every entry point here
accepts the random tensors explicitly (``stratified_u=``, ``cell_jitter=``) so that parity tests can
feed identical numbers to the oracle; when omitted they are drawn with torch on the device.
"""
from __future__ import annotations

from enum import Enum
from typing import Callable, List, Optional, Union

import torch
import torch.nn as nn

from .. import _lib as L
from .. import ops
from ..ops import (accumulate_along_rays, pack_info,  # noqa: F401
                   render_transmittance_from_alpha, render_visibility, render_weight_from_alpha,
                   unpack_info)
from ..ops import ray_aabb_intersect as _ray_aabb_intersect_1box


def ray_aabb_intersect(rays_o, rays_d, aabbs, near_plane: float = -float("inf"),
                       far_plane: float = float("inf"), miss_value: float = float("inf")):
    """nerfacc 0.5.3 form, as called at models/neus.py:164:
    ``ray_aabb_intersect(rays_o, rays_d, aabbs [M,6]) -> (t_mins [N,M], t_maxs [N,M], hits bool [N,M])``.
    One slab-test kernel launch per box (M = 1 at the call site).  The slab test itself is the vendored
    0.3.5 kernel's (lib/nerfacc/cuda/csrc/intersection.cu:16-66: t_min clamped to >= 0, miss = 1e10), so
    ``hits`` = "t_min is not the miss marker" and misses are rewritten to ``miss_value``."""
    aabbs = torch.as_tensor(aabbs, dtype=torch.float32, device=rays_o.device).reshape(-1, 6)
    t_mins, t_maxs = [], []
    for m in range(aabbs.shape[0]):
        tn, tf = _ray_aabb_intersect_1box(rays_o, rays_d, aabbs[m].contiguous())
        t_mins.append(tn)
        t_maxs.append(tf)
    t_mins, t_maxs = torch.stack(t_mins, -1), torch.stack(t_maxs, -1)
    hits = t_mins < 1e10
    t_mins = torch.clamp(t_mins, min=near_plane, max=far_plane)
    t_maxs = torch.clamp(t_maxs, min=near_plane, max=far_plane)
    hits = hits & (t_maxs > t_mins)
    t_mins = torch.where(hits, t_mins, torch.full_like(t_mins, miss_value))
    t_maxs = torch.where(hits, t_maxs, torch.full_like(t_maxs, miss_value))
    return t_mins, t_maxs, hits


def render_weight_from_density(t_starts, t_ends, sigmas, packed_info=None, ray_indices=None, n_rays=None,
                               prefix_trans=None):
    """nerfacc 0.5.3 (models/neus.py:195-197, models/volrend.py:227-233): flat ``[S]`` inputs ->
    ``(weights, trans, alphas)`` with ``alpha = 1 - exp(-sigma (t_end - t_start))`` and the same per-ray scan
    as :func:`render_weight_from_alpha` (lib/nerfacc/vol_rendering.py:201-262 has the 0.3.5 wording).
    Only the learned-background branch of the reference uses densities (disabled in both shipped configs), so
    the density -> alpha step is three torch elementwise ops in front of the HIP scan."""
    if prefix_trans is not None:
        raise NotImplementedError("prefix_trans is not used by RISE-SDF")
    alphas = 1.0 - torch.exp(-sigmas * (t_ends - t_starts))
    weights, trans = render_weight_from_alpha(alphas, packed_info=packed_info, ray_indices=ray_indices,
                                              n_rays=n_rays)
    return weights, trans, alphas


class ContractionType(Enum):
    """lib/nerfacc/contraction.py:13-64 (only AABB is on the hot path)."""
    AABB = 0
    UN_BOUNDED_TANH = 1
    UN_BOUNDED_SPHERE = 2


def _meshgrid3d(res):
    rx, ry, rz = [int(r) for r in res]
    return torch.stack(torch.meshgrid(torch.arange(rx), torch.arange(ry), torch.arange(rz),
                                      indexing="ij"), dim=-1)


class OccupancyGrid(nn.Module):
    """lib/nerfacc/grid.py:113-294.  State: ``occs`` fp32 [cells], ``_binary`` bool [rx,ry,rz]."""

    NUM_DIM = 3

    def __init__(self, roi_aabb, resolution: Union[int, List[int], torch.Tensor] = 128,
                 contraction_type: ContractionType = ContractionType.AABB):
        super().__init__()
        if contraction_type != ContractionType.AABB:
            raise NotImplementedError("only ContractionType.AABB is on the RISE-SDF hot path")
        if isinstance(resolution, int):
            resolution = [resolution] * 3
        resolution = torch.as_tensor(resolution, dtype=torch.int32)
        roi_aabb = torch.as_tensor(roi_aabb, dtype=torch.float32)
        assert resolution.shape == (3,) and roi_aabb.shape == (6,)
        self.num_cells = int(resolution.prod().item())
        self._contraction_type = contraction_type
        self.register_buffer("_roi_aabb", roi_aabb)
        self.register_buffer("_binary", torch.zeros(resolution.tolist(), dtype=torch.bool))
        self.register_buffer("resolution", resolution)
        self.register_buffer("occs", torch.zeros(self.num_cells))
        self.register_buffer("grid_coords", _meshgrid3d(resolution).reshape(self.num_cells, 3),
                             persistent=False)
        self.register_buffer("grid_indices", torch.arange(self.num_cells), persistent=False)

    @property
    def device(self):
        return self._roi_aabb.device

    @property
    def roi_aabb(self):
        return self._roi_aabb

    @property
    def binary(self):
        return self._binary

    @property
    def contraction_type(self):
        return self._contraction_type

    @torch.no_grad()
    def _sample_uniform_and_occupied_cells(self, n: int) -> torch.Tensor:
        uniform = torch.randint(self.num_cells, (n,), device=self.device)
        occupied = torch.nonzero(self._binary.flatten())[:, 0]
        if n < len(occupied):
            occupied = occupied[torch.randint(len(occupied), (n,), device=self.device)]
        return torch.cat([uniform, occupied], dim=0)

    @torch.no_grad()
    def _update(self, step: int, occ_eval_fn: Callable, occ_thre: float = 0.01,
                ema_decay: float = 0.95, warmup_steps: int = 256, indices=None, cell_jitter=None):
        """EMA update (grid.py:196-239).  ``indices`` / ``cell_jitter`` may be supplied for
        reproducible tests; otherwise they are drawn as the reference does."""
        if indices is None:
            indices = self.grid_indices if step < warmup_steps else \
                self._sample_uniform_and_occupied_cells(self.num_cells // 4)
        all_cells = indices is self.grid_indices
        if cell_jitter is None:
            cell_jitter = torch.rand((indices.numel(), 3), dtype=torch.float32, device=self.occs.device)
        # cell -> world point (AABB inverse contraction, helpers_contraction.h:23-28), EMA + threshold: HIP
        x = ops.occ_cell_points(None if all_cells else indices, cell_jitter, self._roi_aabb, self.resolution.tolist())
        occ = occ_eval_fn(x).squeeze(-1)
        ops.occ_update(self.occs, self._binary.view(torch.uint8).view(-1), None if all_cells else indices, occ,
                       ema_decay, occ_thre)

    @torch.no_grad()
    def every_n_step(self, step: int, occ_eval_fn: Callable, occ_thre: float = 1e-2,
                     ema_decay: float = 0.95, warmup_steps: int = 256, n: int = 16):
        if not self.training:
            raise RuntimeError("every_n_step() is a training-time call; use _update() directly "
                               "during inference")
        if step % n == 0:
            self._update(step, occ_eval_fn, occ_thre, ema_decay, warmup_steps)

    @torch.no_grad()
    def query_occ(self, samples: torch.Tensor) -> torch.Tensor:
        return ops.query_occ(samples, self._roi_aabb, self._binary)


@torch.no_grad()
def ray_marching(rays_o, rays_d, t_min=None, t_max=None, scene_aabb=None, grid=None,
                 sigma_fn=None, alpha_fn=None, early_stop_eps: float = 1e-4,
                 alpha_thre: float = 0.0, near_plane=None, far_plane=None,
                 render_step_size: float = 1e-3, stratified: bool = False,
                 cone_angle: float = 0.0, stratified_u: Optional[torch.Tensor] = None,
                 return_packed_info: bool = False):
    """lib/nerfacc/ray_marching.py:13-222 (same argument meaning and error behaviour).

    Returns (ray_indices int64 [S], t_starts [S,1], t_ends [S,1]) like the vendored 0.3.5 API.
    """
    if not rays_o.is_cuda:
        raise NotImplementedError("Only support cuda inputs.")
    if alpha_fn is not None and sigma_fn is not None:
        raise ValueError("Only one of `alpha_fn` and `sigma_fn` should be provided.")
    if t_min is None or t_max is None:
        if scene_aabb is not None:
            t_min, t_max = _ray_aabb_intersect_1box(rays_o, rays_d, scene_aabb)
        else:
            t_min = torch.zeros_like(rays_o[..., 0])
            t_max = torch.ones_like(rays_o[..., 0]) * 1e10
    if near_plane is not None:
        t_min = torch.clamp(t_min, min=near_plane)
    if far_plane is not None:
        t_max = torch.clamp(t_max, max=far_plane)
    if stratified or stratified_u is not None:
        u = torch.rand_like(t_min) if stratified_u is None else stratified_u.to(t_min)
        t_min = t_min + u * render_step_size
    if grid is not None:
        roi, binary = grid.roi_aabb, grid.binary
    else:
        roi = torch.tensor([-1e10, -1e10, -1e10, 1e10, 1e10, 1e10], dtype=torch.float32,
                           device=rays_o.device)
        binary = torch.ones([1, 1, 1], dtype=torch.bool, device=rays_o.device)
    packed_info, ray_indices, t_starts, t_ends = ops.march(
        rays_o, rays_d, t_min, t_max, roi, binary, render_step_size, cone_angle)

    if sigma_fn is not None or alpha_fn is not None:
        ts1, te1 = t_starts[:, None], t_ends[:, None]
        if sigma_fn is not None:
            sigmas = sigma_fn(ts1, te1, ray_indices)
            assert sigmas.shape == ts1.shape, f"sigmas must have shape of (N, 1)! Got {sigmas.shape}"
            alphas = 1.0 - torch.exp(-sigmas * (te1 - ts1))
        else:
            alphas = alpha_fn(ts1, te1, ray_indices)
            assert alphas.shape == ts1.shape, f"alphas must have shape of (N, 1)! Got {alphas.shape}"
        keep = render_visibility(alphas.reshape(-1), packed_info=packed_info,
                                 early_stop_eps=early_stop_eps, alpha_thre=alpha_thre)
        ray_indices, t_starts, t_ends = ops.compact_samples(keep, ray_indices, t_starts, t_ends)
        packed_info = None
    out = (ray_indices, t_starts[:, None], t_ends[:, None])
    return out + (packed_info,) if return_packed_info else out


class OccGridEstimator(nn.Module):
    """nerfacc 0.5.3 estimator surface used at models/split_mixed_occ.py:83-86,126-131,200-208,
    264-272 and models/neus.py:75-78,122,231-238.

    Buffers keep the 0.5.3 names/shapes (one level): ``aabbs`` [1,6], ``occs`` [cells],
    ``binaries`` bool [1,rx,ry,rz], ``resolution`` [3].
    """

    DIM = 3

    def __init__(self, roi_aabb, resolution: Union[int, List[int], torch.Tensor] = 128,
                 levels: int = 1):
        super().__init__()
        if levels != 1:
            raise NotImplementedError("RISE-SDF uses a single-level grid")
        if isinstance(resolution, int):
            resolution = [resolution] * 3
        resolution = torch.as_tensor(resolution, dtype=torch.int32)
        roi_aabb = torch.as_tensor(roi_aabb, dtype=torch.float32)
        self.levels = 1
        self.cells_per_lvl = int(resolution.prod().item())
        self.register_buffer("resolution", resolution)
        self.register_buffer("aabbs", roi_aabb[None, :].clone())
        # the ROI's diagonal, known on the host: the longest [t_min, t_max) of a unit-direction ray, which sizes the staged
        # marcher's per-ray slots (ops.march; a performance hint -- rays that need more are marched twice, as the reference)
        self._t_range_hint = float((roi_aabb[3:] - roi_aabb[:3]).norm())
        self.register_buffer("occs", torch.zeros(self.cells_per_lvl))
        self.register_buffer("binaries", torch.zeros([1] + resolution.tolist(), dtype=torch.bool))
        self.register_buffer("grid_coords", _meshgrid3d(resolution).reshape(self.cells_per_lvl, 3),
                             persistent=False)
        self.register_buffer("grid_indices", torch.arange(self.cells_per_lvl), persistent=False)
        # optional torch.Generator for the update's random cells / jitter (rise_sdf_amd.step.TrainStep seeds one alike on
        # every rank so that the replicated grids stay identical); None = the global generator, as the reference
        self.rng = None
        self.capacity_mode = False
        self._capacity = {}
        self._pending = []
        self._blind_rays = {}            # ray count each read-free capacity was measured at
        self._blind_stale = set()        # blind keys that take the exact path once (grid updated / their last pass overflowed)
        self._blind_all_stale = False    # ... every key, including ones not seen yet (set by a grid update)
        self._blind_seen = set()
        self.stats = {"capped_calls": 0, "overflows": 0, "blind_calls": 0, "blind_overflows": 0,
                      "last_blind_overflow": None}     # (key, candidates, capacity) of the last truncated read-free pass

    @property
    def device(self):
        return self.aabbs.device

    @torch.no_grad()
    def sampling(self, rays_o, rays_d, sigma_fn=None, alpha_fn=None, near_plane: float = 0.0,
                 far_plane: float = 1e10, t_min=None, t_max=None, render_step_size: float = 1e-3,
                 early_stop_eps: float = 1e-4, alpha_thre: float = 0.0, stratified: bool = False,
                 cone_angle: float = 0.0, stratified_u: Optional[torch.Tensor] = None, return_alphas: bool = False):
        """-> (ray_indices int64 [S], t_starts [S], t_ends [S]); ``alpha_fn(t_starts, t_ends,
        ray_indices) -> [S]`` prunes by visibility exactly like the vendored marcher does.  ``return_alphas`` (with
        ``alpha_fn``): a fourth output, the kept samples' alphas as ``alpha_fn`` returned them for the visibility test -- a
        no-grad renderer of these samples would evaluate the field a second time for the same values."""
        if return_alphas and alpha_fn is None:
            raise ValueError("OccGridEstimator.sampling(return_alphas=True) needs alpha_fn: the fourth output is what alpha_fn "
                             "returned for the kept samples (with sigma_fn or no fn there is nothing to hand back)")
        near = None if near_plane is None else float(near_plane)
        far = None if far_plane is None else float(far_plane)
        if t_min is None or t_max is None:
            t_min, t_max = _ray_aabb_intersect_1box(rays_o, rays_d, self.aabbs[0])
        if near is not None:
            t_min = torch.clamp(t_min, min=near)
        if far is not None:
            t_max = torch.clamp(t_max, max=far)
        if stratified or stratified_u is not None:
            u = torch.rand_like(t_min) if stratified_u is None else stratified_u.to(t_min)
            t_min = t_min + u * render_step_size
        if self.capacity_mode and alpha_fn is not None and sigma_fn is None:
            out = self._sampling_capped(rays_o, rays_d, t_min, t_max, alpha_fn, render_step_size, cone_angle,
                                        early_stop_eps, alpha_thre, key=(near, far, round(float(render_step_size), 9)),
                                        return_alphas=return_alphas)
            if out is not None:
                return out
        packed_info, ray_indices, t_starts, t_ends = ops.march(
            rays_o, rays_d, t_min, t_max, self.aabbs[0], self.binaries[0], render_step_size,
            cone_angle, t_range_hint=self._t_range_hint)
        if self.capacity_mode:
            if self._pending:        # (an exact call -- first step, or after an overflow -- settles them with a read of its own)
                vals = torch.cat([p[2] for p in self._pending]).tolist()
                self._settle_pending(list(zip(vals[0::2], vals[1::2])))
            self._remember(key=(near, far, round(float(render_step_size), 9)), n_candidates=ray_indices.numel())
            bkey = ("blind", near, far, round(float(render_step_size), 9))
            if bkey not in self._capacity:
                self._capacity[bkey] = int(ray_indices.numel() * 1.5) + 4096
                self._blind_rays[bkey] = int(rays_o.shape[0])
        if (sigma_fn is not None or alpha_fn is not None) and ray_indices.numel() > 0:
            if sigma_fn is not None:
                sigmas = sigma_fn(t_starts, t_ends, ray_indices)
                alphas = 1.0 - torch.exp(-sigmas * (t_ends - t_starts))
            else:
                alphas = alpha_fn(t_starts, t_ends, ray_indices)
            assert alphas.shape == t_starts.shape, \
                f"alphas must have shape of (N,)! Got {alphas.shape}"
            for _attempt in range(3):
                n_reroutes = L.status_totals()["range_reroutes"]
                keep = render_visibility(alphas, packed_info=packed_info,
                                         early_stop_eps=early_stop_eps, alpha_thre=alpha_thre)
                res = ops.compact_samples(keep, ray_indices, t_starts, t_ends,
                                          extra=alphas if (return_alphas and alpha_fn is not None) else None)
                if L.status_totals()["range_reroutes"] == n_reroutes or sigma_fn is not None:
                    break
                # the poll behind the compaction's host read found that alpha_fn left the range of the two-part fp16
                # kernels and switched them off (_lib.poll_status): these alphas are not the reference's -- again, on the
                # range-free kernels (models/network_utils.py:109-157: the reference's fp32 MLP just continues)
                alphas = alpha_fn(t_starts, t_ends, ray_indices)
            if return_alphas and alpha_fn is not None:
                return res
            ray_indices, t_starts, t_ends = res[:3]
        elif return_alphas:
            return ray_indices, t_starts, t_ends, torch.zeros_like(t_starts)      # (alpha_fn given, but no samples at all)
        return ray_indices, t_starts, t_ends

    # ---- capacity mode (opt-in: ``estimator.capacity_mode = True``; rise_sdf_amd.step.TrainStep switches it on) ----------
    # Visibility-pruned sampling reads two counts back per call in the reference (the marcher's candidate total,
    # ray_marching.cu:257-261, and the survivors after the boolean-mask compaction).  With buffers sized from the previous
    # call of the same kind (+15 %) the marcher, alpha_fn, the visibility test and the compaction are all enqueued without
    # knowing either count, and both are read together afterwards: ONE host read per sampling call.  The samples, their
    # order and every value are identical to the exact path (tests/test_gpu_edges.py); if the candidates outgrew the
    # buffers the call is redone the exact way and the capacity grows.
    def _remember(self, key, n_candidates):
        self._capacity[key] = int(n_candidates * 1.15) + 4096

    def _sampling_capped(self, rays_o, rays_d, t_min, t_max, alpha_fn, render_step_size, cone_angle, early_stop_eps,
                         alpha_thre, key, return_alphas=False):
        cap = self._capacity.get(key)
        if cap is None:
            return None
        packed, ri, ts, te, total = ops.march_capped(rays_o, rays_d, t_min, t_max, self.aabbs[0], self.binaries[0],
                                                     render_step_size, cap, cone_angle, t_range_hint=self._t_range_hint)
        alphas = alpha_fn(ts, te, ri)
        keep = render_visibility(alphas, packed_info=packed, early_stop_eps=early_stop_eps, alpha_thre=alpha_thre,
                                 zero_init=True)
        cnt = []
        res = ops.compact_samples(keep, ri, ts, te, count_out=cnt, extra=alphas if return_alphas else None)
        ri_o, ts_o, te_o = res[:3]
        st = L.status(total.device)          # (the kernels' sticky status words ride along with the counts)
        vals = torch.cat([total, cnt[0]] + [p[2] for p in self._pending] + [st]).tolist()   # the one host read of this call
        vals, st_vals = vals[:-L.STATUS_WORDS], vals[-L.STATUS_WORDS:]
        rerouted = L.consume_status(st, st_vals)["rerouted_now"]
        n_cand, n_kept = vals[:2]
        self._settle_pending(list(zip(vals[2::2], vals[3::2])))      # (+ the counts of earlier read-free passes)
        self._remember(key, n_cand)
        bkey = ("blind",) + tuple(key)
        if bkey not in self._capacity:       # (dropped by sampling_blind when it sent its caller here: re-measured now)
            self._capacity[bkey] = int(n_cand * 1.5) + 4096
            self._blind_rays[bkey] = int(rays_o.shape[0])
        self.stats["capped_calls"] += 1
        if n_cand > cap:                                               # truncated: redo exactly (rare)
            self.stats["overflows"] += 1
            return None
        if rerouted:         # alpha_fn left the x2 kernels' range (they are switched off now): redo on the range-free kernels
            return None
        if return_alphas:
            return ri_o[:n_kept], ts_o[:n_kept], te_o[:n_kept], res[3][:n_kept]
        return ri_o[:n_kept], ts_o[:n_kept], te_o[:n_kept]

    @torch.no_grad()
    def sampling_blind(self, rays_o, rays_d, alpha_fn, near_plane, far_plane, t_min, t_max, render_step_size,
                       early_stop_eps: float = 1e-4, alpha_thre: float = 0.0, return_alphas: bool = False):
        """Capacity mode WITHOUT a host read, for passes whose results are per-ray only (the secondary-ray occlusion
        pass): -> (ray_indices, t_starts, t_ends), all of capacity size; entries past the survivor count are empty samples
        of the LAST ray (the caller appends a phantom ray that owns them), or None when no capacity is known yet.  The
        two counts stay on the device and ride along with the next capped call's read; if the candidates outgrew the
        buffers, that pass ran on a truncated set (``stats['blind_overflows']``) and the capacity grows."""
        key = ("blind", float(near_plane), float(far_plane), round(float(render_step_size), 9))
        cap = self._capacity.get(key)
        t_min = torch.clamp(t_min, min=float(near_plane))
        t_max = torch.clamp(t_max, max=float(far_plane))
        if self._blind_all_stale:          # a grid update invalidates EVERY key's capacity, not only the next caller's
            self._blind_stale |= self._blind_seen
            self._blind_all_stale = False
        self._blind_seen.add(key)
        if cap is None or key in self._blind_stale:
            # no capacity yet, the grid has just been updated, or this key's last blind pass overflowed: the caller takes
            # the exact (read-ful) path once, which also re-measures the capacity
            self._blind_stale.discard(key)
            self._capacity.pop(key, None)
            return None
        # the capacity was measured for another ray count (dynamic_ray_sampling ramps train_num_rays by up to ~3x per
        # step early on): scale it by the host-known ratio
        n_rays = int(rays_o.shape[0])
        n_meas = self._blind_rays.get(key, n_rays)
        if n_rays > n_meas + max(8, n_meas // 20):      # (the caller's phantom ray alone is not "more rays")
            cap = int(cap * (n_rays / max(n_meas, 1))) + 4096
        packed, ri, ts, te, total = ops.march_capped(rays_o, rays_d, t_min, t_max, self.aabbs[0], self.binaries[0],
                                                     render_step_size, cap, 0.0, t_range_hint=self._t_range_hint)
        alphas = alpha_fn(ts, te, ri)
        keep = render_visibility(alphas, packed_info=packed, early_stop_eps=early_stop_eps, alpha_thre=alpha_thre,
                                 zero_init=True)
        cnt = []
        out = ops.compact_samples(keep, ri, ts, te, count_out=cnt, fill_ray=rays_o.shape[0] - 1,
                                  extra=alphas if return_alphas else None)       # (+ the kept samples' alphas, tail zeros)
        self._pending.append((key, cap, torch.cat([total, cnt[0]]), n_rays))
        self.stats["blind_calls"] += 1
        return out

    def _settle_pending(self, counts):
        """counts: the host values of the pending blind calls' [candidates, survivors], in order."""
        for p, (n_cand, _) in zip(self._pending, counts):
            key, cap = p[0], p[1]
            self._capacity[key] = int(n_cand * 1.5) + 4096
            if len(p) > 3:
                self._blind_rays[key] = p[3]
            if n_cand > cap:
                # that pass ran on a truncated sample set (its tail rays saw transmittance 1): it cannot be redone after
                # the fact, so say so and take the exact path next time
                self.stats["blind_overflows"] += 1
                self.stats["last_blind_overflow"] = (key, int(n_cand), int(cap))     # (TrainStep reports it with the step)
                self._blind_stale.add(key)
                import warnings
                warnings.warn(f"rise_sdf_amd.nerfacc: a read-free secondary sampling pass outgrew its buffers "
                              f"({n_cand} candidates > capacity {cap}); its tail rays were left unoccluded for that one "
                              f"step.  The next pass takes the exact path.", RuntimeWarning, stacklevel=2)
        self._pending = []

    @torch.no_grad()
    def _sample_uniform_and_occupied_cells(self, n: int):
        uniform = torch.randint(self.cells_per_lvl, (n,), device=self.device, generator=self.rng)
        occupied = torch.nonzero(self.binaries[0].flatten())[:, 0]
        if n < len(occupied):
            occupied = occupied[torch.randint(len(occupied), (n,), device=self.device, generator=self.rng)]
        return torch.cat([uniform, occupied], dim=0)

    @torch.no_grad()
    def _update(self, step: int, occ_eval_fn: Callable, occ_thre: float = 0.01,
                ema_decay: float = 0.95, warmup_steps: int = 256, indices=None, cell_jitter=None):
        if indices is None:
            indices = self.grid_indices if step < warmup_steps else \
                self._sample_uniform_and_occupied_cells(self.cells_per_lvl // 4)
        all_cells = indices is self.grid_indices
        if cell_jitter is None:
            cell_jitter = torch.rand((indices.numel(), 3), dtype=torch.float32, device=self.occs.device,
                                     generator=self.rng)
        x = ops.occ_cell_points(None if all_cells else indices, cell_jitter, self.aabbs[0], self.resolution.tolist())
        occ = occ_eval_fn(x).squeeze(-1)
        ops.occ_update(self.occs, self.binaries.view(torch.uint8).view(-1), None if all_cells else indices, occ,
                       ema_decay, occ_thre)
        self._blind_all_stale = True       # the occupied set has changed: every read-free pass re-measures its capacity

    @torch.no_grad()
    def update_every_n_steps(self, step: int, occ_eval_fn: Callable, occ_thre: float = 1e-2,
                             ema_decay: float = 0.95, warmup_steps: int = 256, n: int = 16):
        if not self.training:
            raise RuntimeError("update_every_n_steps() is a training-time call")
        if step % n == 0:
            self._update(step, occ_eval_fn, occ_thre, ema_decay, warmup_steps)


from . import volrend  # noqa: E402,F401  (``from nerfacc.volrend import ...``, models/volrend.py:10-14)
