"""Host-side mirror of the reference's NeuS geometry renderer.

  VarianceNetwork   models/neus.py:21-49 (== models/split_mixed_occ.py:21-56)
  NeuSModel         models/neus.py:52-351 (``neus``), with the finite-difference-normal
                    ``alpha_fn`` of models/split_mixed_occ.py:228-240 and the per-ray
                    compositing of models/volrend.py:851-886.

Without a ``texture`` node and with finite-difference normals this is BASELINE.json config[1]: hash-grid SDF +
NeuS alpha + transmittance compositing of opacity / depth / normals, forward and backward, on the fused
stencil kernels.  With a ``texture`` (``volume-radiance``) and / or ``grad_type: analytic`` (neus-blender.yaml) it
takes the general path of models/neus.py:240-317 -- geometry(positions) -> get_alpha -> texture -> composite -- and
returns ``comp_rgb`` plus the ``_bg`` / ``_full`` dictionaries.  The learned-background branch is disabled in both
shipped configs and not built.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .geometry import BaseModel
from .nerfacc import OccGridEstimator
from .network_utils import update_module_step
from .registry import make, register


class VarianceNetwork(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.init_val = self.config.init_val
        self.register_parameter("variance", nn.Parameter(torch.tensor(float(self.config.init_val))))
        self.modulate = self.config.get("modulate", False)
        if self.modulate:
            self.mod_start_steps = self.config.mod_start_steps
            self.reach_max_steps = self.config.reach_max_steps
            self.max_inv_s = self.config.max_inv_s
            self.do_mod = False

    @property
    def inv_s(self):
        val = torch.exp(self.variance * 10.0)
        if self.modulate and self.do_mod:
            val = val.clamp_max(self.mod_val)
        return val

    def forward(self, x):
        return torch.ones([len(x), 1], device=self.variance.device) * self.inv_s

    def effective_variance(self):
        """The scalar the alpha kernels exponentiate (inv_s = clip(exp(10 v), 1e-6, 1e6) on the device).  With
        ``modulate`` active (models/neus.py:36-37: ``inv_s.clamp_max(mod_val)``) the clamp is applied to v instead:
        exp(10 min(v, ln(mod_val) / 10)) == min(exp(10 v), mod_val) up to rounding, and ``torch.minimum`` gives the
        clamp's gradient (none while the clamp is active)."""
        if self.modulate and self.do_mod:
            import math
            return torch.minimum(self.variance, self.variance.new_tensor(math.log(self.mod_val) / 10.0))
        return self.variance

    def update_step(self, epoch, global_step):
        if self.modulate:
            self.do_mod = global_step > self.mod_start_steps
            if not self.do_mod:
                self.prev_inv_s = self.inv_s.item()
            else:
                self.mod_val = min((global_step / self.reach_max_steps)
                                   * (self.max_inv_s - self.prev_inv_s) + self.prev_inv_s,
                                   self.max_inv_s)


_EVAL_STREAMS = {}


def chunk_batch(func, chunk_size, move_to_cpu, *args, streams=1, **kwargs):
    """models/utils.py:14-51 for dict-returning functions.

    ``streams`` > 1 (evaluation only: no autograd graph is kept): consecutive chunks are issued on alternating HIP streams.
    An evaluation chunk blocks the host several times (sample counts), and while one chunk waits the other stream's kernels
    keep the GPU busy; results are identical (every chunk's kernels stay in order on their own stream) and everything is
    joined to the caller's stream before the outputs are concatenated."""
    B = next(a.shape[0] for a in args if isinstance(a, torch.Tensor))
    dev = next((a.device for a in args if isinstance(a, torch.Tensor)), None)
    use = (streams > 1 and dev is not None and dev.type == "cuda" and not torch.is_grad_enabled() and B > chunk_size)
    pool = None
    if use:
        key = (str(dev), int(streams))
        if key not in _EVAL_STREAMS:
            _EVAL_STREAMS[key] = [torch.cuda.Stream(device=dev) for _ in range(int(streams))]
        pool, main = _EVAL_STREAMS[key], torch.cuda.current_stream(dev)
        for st in pool:
            st.wait_stream(main)
    out = {}
    for k, i in enumerate(range(0, B, chunk_size)):
        sl = [a[i:i + chunk_size] if isinstance(a, torch.Tensor) else a for a in args]
        if pool is not None:
            with torch.cuda.stream(pool[k % len(pool)]):
                chunk = func(*sl, **kwargs)
                for v in chunk.values():
                    if isinstance(v, torch.Tensor):
                        v.record_stream(main)          # consumed (torch.cat) on the caller's stream below
        else:
            chunk = func(*sl, **kwargs)
        for k2, v in chunk.items():
            v = v if torch.is_grad_enabled() else v.detach()
            out.setdefault(k2, []).append(v.cpu() if move_to_cpu else v)
    if pool is not None:
        for st in pool:
            main.wait_stream(st)
    return {k: torch.cat(v, dim=0) for k, v in out.items()}


@register("neus")
class NeuSModel(BaseModel):
    def setup(self):
        self.geometry = make(self.config.geometry.name, self.config.geometry)
        tex = self.config.get("texture", None)
        self.texture = make(tex.name, tex) if tex is not None else None
        if self.config.get("learned_background", False):
            raise NotImplementedError("learned_background is disabled in both shipped configs")
        self.variance = VarianceNetwork(self.config.variance)
        # fused stencil kernels: SDF-only rendering with finite-difference normals.  A radiance network or analytic
        # normals (neus-blender.yaml) take the general path: geometry(...) -> get_alpha -> texture -> composite.
        self._general = self.texture is not None or self.geometry.grad_type != "finite_difference"
        r = float(self.config.radius)
        self.register_buffer("scene_aabb", torch.tensor([-r, -r, -r, r, r, r], dtype=torch.float32))
        self.grid_prune = bool(self.config.get("grid_prune", True))
        self.occupancy_grid = OccGridEstimator(roi_aabb=self.scene_aabb, resolution=128)
        if not self.grid_prune:
            # dense marching: every cell occupied (config[0]/[1] of BASELINE.json)
            self.occupancy_grid.binaries.fill_(True)
        self.randomized = self.config.get("randomized", True)
        self.background_color = None
        self.render_step_size = 1.732 * 2 * self.config.radius / self.config.num_samples_per_ray
        self.cos_anneal_ratio = 1.0
        # split_mixed_occ prunes candidates by visibility (alpha_fn inside sampling); neus.py does not
        self.prune_by_visibility = bool(self.config.get("prune_by_visibility", False))

    # ---- per-step schedule: models/neus.py:93-122 ------------------------------------------------
    def update_step(self, epoch, global_step):
        update_module_step(self.geometry, epoch, global_step)
        if self.texture is not None:
            update_module_step(self.texture, epoch, global_step)
        update_module_step(self.variance, epoch, global_step)
        cos_anneal_end = self.config.get("cos_anneal_end", 0)
        self.cos_anneal_ratio = 1.0 if cos_anneal_end == 0 else min(1.0, global_step / cos_anneal_end)
        if self.training and self.grid_prune:
            self.occupancy_grid.update_every_n_steps(
                step=global_step, occ_eval_fn=self.occ_eval_fn,
                occ_thre=self.config.get("grid_prune_occ_thre", 0.01))

    def occ_eval_fn(self, x):
        """models/neus.py:101-111 (A2): alpha with cos == -1 and delta == render_step_size."""
        sdf = self.geometry(x, with_grad=False, with_feature=False)
        return ops.occ_alpha(sdf, self.variance.effective_variance(), self.render_step_size)      # one kernel (A2)

    def get_alpha(self, sdf, normal, dirs, dists):
        """models/neus.py:128-150."""
        return ops.neus_alpha(sdf, normal, dirs, dists, self.variance.effective_variance(),
                              self.cos_anneal_ratio)

    # ---- one ray batch ------------------------------------------------------------------------------
    def _field7(self, rays_o, rays_d, ray_indices, t_starts, t_ends):
        """-> (stencil, tap_major): the SDF at each sample's centre and six FD taps, either the
        tap-major [7, S] array of the fused stencil kernels or rows 7i+t (column 0) of the per-layer
        path's [7S, feature_dim] output (``fused: false`` in the model config forces the latter)."""
        if self._fused_ok():
            return self.geometry.sdf7_from_rays(rays_o, rays_d, ray_indices, t_starts, t_ends)[0], True
        return self.geometry.field7_from_rays(rays_o, rays_d, ray_indices, t_starts, t_ends), False

    def _fused_ok(self):
        return (not self._general and self.config.get("fused", True) and self.geometry.fused_field_available())

    def _alpha_fn(self, rays_o, rays_d):
        def alpha_fn(t_starts, t_ends, ray_indices):
            if ray_indices.numel() == 0:
                return torch.zeros((0,), device=rays_o.device)
            with torch.no_grad():
                if self.geometry.grad_type != "finite_difference":
                    t_dirs = rays_d[ray_indices]
                    positions = rays_o[ray_indices] + t_dirs * (t_starts + t_ends)[..., None] / 2.0
                    sdf, sdf_grad = self.geometry(positions, with_grad=True, with_feature=False)
                    return self.get_alpha(sdf, F.normalize(sdf_grad, p=2, dim=-1), t_dirs,
                                          (t_ends - t_starts)[..., None])
                out7, tm = self._field7(rays_o, rays_d, ray_indices, t_starts, t_ends)
                return ops.neus_alpha_fd(out7, self.variance.effective_variance(), rays_d, ray_indices,
                                         t_starts, t_ends, self.cos_anneal_ratio,
                                         self.geometry._finite_difference_eps, tap_major=tm)[3]
        return alpha_fn

    def forward_(self, rays, stratified_u=None):
        n_rays = rays.shape[0]
        rays_o, rays_d = rays[:, 0:3].contiguous(), rays[:, 3:6].contiguous()
        with torch.no_grad():
            ray_indices, t_starts, t_ends = self.occupancy_grid.sampling(
                rays_o, rays_d,
                alpha_fn=self._alpha_fn(rays_o, rays_d) if self.prune_by_visibility else None,
                render_step_size=self.render_step_size,
                stratified=self.randomized and stratified_u is None, stratified_u=stratified_u,
                cone_angle=0.0, alpha_thre=0.0)
        return self.render_samples(rays_o, rays_d, ray_indices, t_starts, t_ends, n_rays)

    def render_samples(self, rays_o, rays_d, ray_indices, t_starts, t_ends, n_rays):
        """Field query -> NeuS alpha -> composite for a marched sample set (models/neus.py:240-317,
        models/volrend.py:851-886)."""
        dev = rays_o.device
        S = ray_indices.numel()
        packed = ops.pack_info(ray_indices, n_rays)
        rgb = None
        if S == 0:
            alpha = torch.zeros((0,), device=dev)
            sdf, sdf_grad = torch.zeros((0,), device=dev), torch.zeros((0, 3), device=dev)
            normal = torch.zeros((0, 3), device=dev)
            rgb = torch.zeros((0, 3), device=dev) if self.texture is not None else None
        elif self._general:
            # models/neus.py:250-262, one call each: field (FD or analytic gradient), alpha, radiance
            t_dirs = rays_d[ray_indices]
            positions = rays_o[ray_indices] + t_dirs * (t_starts + t_ends)[..., None] / 2.0
            sdf, sdf_grad, feature = self.geometry(positions, with_grad=True, with_feature=True)
            normal = F.normalize(sdf_grad, p=2, dim=-1)
            alpha = self.get_alpha(sdf, normal, t_dirs, (t_ends - t_starts)[..., None])
            if self.texture is not None:
                rgb = self.texture(feature, t_dirs, normal)
        else:
            out7, tm = self._field7(rays_o, rays_d, ray_indices, t_starts, t_ends)
            sdf, sdf_grad, normal, alpha = ops.neus_alpha_fd(
                out7, self.variance.effective_variance(), rays_d, ray_indices, t_starts, t_ends,
                self.cos_anneal_ratio, self.geometry._finite_difference_eps, tap_major=tm)
        weights, _ = ops.render_weight_from_alpha(alpha, packed_info=packed)
        # opacity, depth (weights . (t_starts + t_ends) / 2) and the normal map in one pass, bit-identical to the three
        # accumulate calls (RSDF_FOLD_NORMALS=0: the normal map through accumulate_along_rays, for A/B and the fold's tests)
        midpoints = None
        if ops.fold_normals():
            res = ops.accumulate_opacity_depth_normal(weights, t_starts, t_ends, normal, packed_info=packed,
                                                      want_midpoints=self.training)
            opacity, depth, comp_normal = res[:3]
            midpoints = res[3] if self.training else None      # (the training outputs carry the sample midpoints, models/neus.py:303)
        else:
            if self.training:
                opacity, depth, midpoints = ops.accumulate_opacity_depth(weights, t_starts, t_ends, packed_info=packed,
                                                                         want_midpoints=True)
            else:
                opacity, depth = ops.accumulate_opacity_depth(weights, t_starts, t_ends, packed_info=packed)
            comp_normal = ops.accumulate_along_rays(weights, normal, packed_info=packed)
        out = {
            "comp_normal_raw": comp_normal,
            "comp_normal": F.normalize(comp_normal, p=2, dim=-1),
            "opacity": opacity,
            "depth": depth,
            "rays_valid": opacity > 0,
            "num_samples": torch.full((1,), S, dtype=torch.int32, device=dev),
        }
        if self.training:
            out["num_samples_host"] = int(S)      # see split_mixed_occ.py: no second host read for dynamic_ray_sampling
            out.update({"sdf_samples": sdf, "sdf_grad_samples": sdf_grad,
                        "weights": weights.view(-1), "points": midpoints.view(-1),
                        "intervals": (t_ends - t_starts).view(-1),
                        "ray_indices": ray_indices.view(-1)})
        if rgb is None:
            return out          # geometry-only model (BASELINE config[1]): no radiance keys
        # models/neus.py:264-317: composited colour + the background / full dictionaries
        out["comp_rgb"] = ops.accumulate_along_rays(weights, rgb, packed_info=packed)
        bg = self.background_color if self.background_color is not None else torch.ones(3, device=dev)
        out_bg = {"comp_rgb": bg[None, :].expand(*out["comp_rgb"].shape),
                  "num_samples": torch.zeros_like(out["num_samples"]),
                  "rays_valid": torch.zeros_like(out["rays_valid"])}
        out_full = {"comp_rgb": out["comp_rgb"] + out_bg["comp_rgb"] * (1.0 - out["opacity"]),
                    "num_samples": out["num_samples"] + out_bg["num_samples"],
                    "rays_valid": out["rays_valid"] | out_bg["rays_valid"]}
        return {**out, **{k + "_bg": v for k, v in out_bg.items()}, **{k + "_full": v for k, v in out_full.items()}}

    def forward(self, rays, **kw):
        if self.training:
            out = self.forward_(rays, **kw)
        else:
            out = chunk_batch(self.forward_, self.config.get("ray_chunk", 4096), False, rays,
                              streams=int(self.config.get("eval_streams", 2)))
        return {**out, "inv_s": self.variance.inv_s}

    def train(self, mode=True):
        self.randomized = mode and self.config.get("randomized", True)
        return super().train(mode=mode)

    def eval(self):
        self.randomized = False
        return super().eval()

    def regularizations(self, out):
        return {}
