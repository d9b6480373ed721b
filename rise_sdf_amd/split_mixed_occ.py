"""Host-side mirror of the reference's ``models/split_mixed_occ.py`` (``split-mixed-occ``).

Built so far: occupancy update (M2, :98-136), visibility-pruned sampling (M4/M5,
:264-272), field + FD normals + NeuS alpha (H1-H4, A1, :228-262), radiance branch at stage 0
(T1-T3, :293-295), compositing (C1-C3, :274-289), secondary-ray occlusion (R1, :179-222,306-318),
normal-orientation map (:383-401), background compose + sRGB (O1, :404-443).
Stage 1 (``split_sum_kick_in_step``): split-sum shading against the prefiltered environment light
(S1-S4, E1; 24 composited channels, :295-303, 344-352, 416-432).
Curvature samples (H5, ``sdf_laplace_samples``) through geometry.curvature.
Relighting (:322-331): third-bounce shading of smooth pixels against a swapped environment (``relight``).
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F

from . import ops
from . import texture_ops as T
from .geometry import BaseModel
from .nerfacc import ContractionType, OccGridEstimator
from .network_utils import update_module_step
from .neus import VarianceNetwork, chunk_batch
from .registry import make, register
from .volrend import rendering_with_normals_sdf, secondary_rendering


@register("split-mixed-occ")
class SplitMixedOCCModel(BaseModel):
    def setup(self):
        self.geometry = make(self.config.geometry.name, self.config.geometry)
        self.texture = make(self.config.texture.name, self.config.texture)
        light = self.config.light if "light" in self.config else None
        self.emitter = make(light.name, light) if light is not None and light.name in _registry() else None
        self.geometry.contraction_type = ContractionType.AABB
        self.variance = VarianceNetwork(self.config.variance)
        r = float(self.config.radius)
        self.register_buffer("scene_aabb", torch.tensor([-r, -r, -r, r, r, r], dtype=torch.float32))
        self.grid_prune = bool(self.config.get("grid_prune", True))
        self.occupancy_grid = OccGridEstimator(roi_aabb=self.scene_aabb, resolution=128)
        if not self.grid_prune:
            self.occupancy_grid.binaries.fill_(True)
        self.randomized = self.config.get("randomized", True)
        self.background_color = None
        self.render_step_size = 1.732 * 2 * self.config.radius / self.config.num_samples_per_ray
        self.num_samples_per_secondary_ray = self.config.get("num_samples_per_secondary_ray", 96)
        self.secondary_near_plane = self.config.get("secondary_near_plane", 0.05)
        self.secondary_far_plane = self.config.get("secondary_far_plane", 1.5)
        self.secondary_shader_chunk = self.config.get("secondary_shader_chunk", 160000)
        self.cos_anneal_ratio = 1.0
        self.stage = 0
        # opt-in (rise_sdf_amd.step.TrainStep): the secondary-ray blend as a select over all rays instead of a
        # gather / scatter over torch.nonzero(opacity > 0.5), which costs a host read per step
        self.masked_secondary = False
        self.after_sampling = None        # one-shot callable run right after the primary sampling of the next forward_

    # ---- per-step schedule (:98-136) ------------------------------------------------------------------
    def update_step(self, epoch, global_step):
        update_module_step(self.geometry, epoch, global_step)
        update_module_step(self.texture, epoch, global_step)
        update_module_step(self.variance, epoch, global_step)
        cos_anneal_end = self.config.get("cos_anneal_end", 0)
        self.cos_anneal_ratio = 1.0 if cos_anneal_end == 0 else min(1.0, global_step / cos_anneal_end)
        if self.training and self.grid_prune:
            self.occupancy_grid.update_every_n_steps(
                step=global_step, occ_eval_fn=self.occ_eval_fn,
                occ_thre=self.config.get("grid_prune_occ_thre", 0.01))
        self.stage = 1 if global_step >= self.config.get("split_sum_kick_in_step", 1 << 62) else 0

    def occ_eval_fn(self, x):
        sdf = self.geometry(x, with_grad=False, with_feature=False)
        return ops.occ_alpha(sdf, self.variance.effective_variance(), self.render_step_size)      # one kernel (A2)

    def get_alpha(self, sdf, normal, dirs, dists):
        return ops.neus_alpha(sdf, normal, dirs, dists, self.variance.effective_variance(), self.cos_anneal_ratio)

    # ---- field access -----------------------------------------------------------------------------------
    def _stencil(self, rays_o, rays_d, ray_indices, t_starts, t_ends, want_feature):
        """-> (sdf, sdf_grad, normal, alpha, feature or None) for a sample set."""
        geo = self.geometry
        eps = geo._finite_difference_eps
        if geo.grad_type != "finite_difference":
            # analytic normals (models/geometry.py:224-228): the general field call + the stand-alone alpha kernel
            t_dirs = rays_d[ray_indices]
            positions = rays_o[ray_indices] + t_dirs * (t_starts + t_ends)[..., None] / 2.0
            sdf, sdf_grad, feature = geo(positions, with_grad=True, with_feature=True)
            normal = F.normalize(sdf_grad, p=2, dim=-1, eps=1e-6)
            alphas = self.get_alpha(sdf, normal, t_dirs, (t_ends - t_starts)[..., None])
            return sdf, sdf_grad, normal, alphas, (feature if want_feature else None)
        if self.config.get("fused", True) and geo.fused_field_available():
            sdf7t, feature = geo.sdf7_from_rays(rays_o, rays_d, ray_indices, t_starts, t_ends,
                                                want_feature=want_feature)
            out = ops.neus_alpha_fd(sdf7t, self.variance.effective_variance(), rays_d, ray_indices, t_starts, t_ends,
                                    self.cos_anneal_ratio, eps, tap_major=True)
            return (*out, feature)
        out7 = geo.field7_from_rays(rays_o, rays_d, ray_indices, t_starts, t_ends)
        out = ops.neus_alpha_fd(out7, self.variance.effective_variance(), rays_d, ray_indices, t_starts, t_ends,
                                self.cos_anneal_ratio, eps)
        feature = out7.view(-1, 7, out7.shape[-1])[:, 0] if want_feature else None
        return (*out, feature)

    def _alpha_fn(self, rays_o, rays_d):
        def alpha_fn(t_starts, t_ends, ray_indices):
            if ray_indices.numel() == 0:
                return torch.zeros((0,), device=rays_o.device)
            with torch.no_grad():
                return self._stencil(rays_o, rays_d, ray_indices, t_starts, t_ends, False)[3]
        return alpha_fn

    # ---- secondary-ray occlusion (R1, :179-222) -------------------------------------------------------------
    def compute_indirect_radiance(self, rays_o, rays_d, valid=None):
        """``valid`` (bool [n_rays], masked-secondary mode): rays outside it are given an empty [t_min, t_max] interval, so
        they march to nothing; the others see exactly the interval the slab test gives them."""
        n_rays = rays_o.shape[0]
        alpha_fn = self._alpha_fn(rays_o, rays_d)
        # The secondary rays are rendered without gradients from the samples the visibility test has just kept: their alphas
        # are the sampler's own (the reference evaluates the field again for them, models/volrend.py:60-75 after
        # lib/nerfacc/ray_marching.py:193-220, and gets the same values; here the same kernels would: bit-identical,
        # tests/test_gpu_split_model.py).  RSDF_SECONDARY_REUSE_ALPHA=0 evaluates twice.
        reuse = os.environ.get("RSDF_SECONDARY_REUSE_ALPHA", "1") != "0"
        with torch.no_grad():
            step = (self.secondary_far_plane - self.secondary_near_plane) / (self.num_samples_per_secondary_ray - 1)
            t_min = t_max = None
            if valid is not None:
                t_min, t_max = ops.ray_aabb_intersect(rays_o, rays_d, self.occupancy_grid.aabbs[0])
                miss = torch.full_like(t_min, 1e10)
                t_min, t_max = torch.where(valid, t_min, miss), torch.where(valid, t_max, miss)
                if self.occupancy_grid.capacity_mode:
                    # No host read at all for this pass (its results are per ray): capacity-sized sample arrays whose
                    # unused tail belongs to one phantom ray appended here; counts are checked with the next step's read.
                    ro_p = torch.cat([rays_o, torch.zeros_like(rays_o[:1])])            # (fills, no host-to-device copies)
                    rd_p = torch.cat([rays_d, torch.ones_like(rays_d[:1]) * 0.57735026])
                    a_fn = self._alpha_fn(ro_p, rd_p)
                    far1 = torch.full_like(t_min[:1], 1e10)
                    blind = self.occupancy_grid.sampling_blind(
                        ro_p, rd_p, a_fn, self.secondary_near_plane, self.secondary_far_plane,
                        torch.cat([t_min, far1]), torch.cat([t_max, far1]), step, return_alphas=reuse)
                    if blind is not None:
                        acc_map, depth_map, _ = secondary_rendering(blind[1], blind[2], ray_indices=blind[0],
                                                                    n_rays=n_rays + 1, alpha_fn=a_fn,
                                                                    chunk_size=self.secondary_shader_chunk,
                                                                    phantom_last_ray=True,
                                                                    alphas=blind[3] if reuse else None)
                        return 1.0 - acc_map[:n_rays], depth_map[:n_rays]
            res = self.occupancy_grid.sampling(
                rays_o, rays_d, alpha_fn=alpha_fn, near_plane=self.secondary_near_plane,
                far_plane=self.secondary_far_plane, render_step_size=step, stratified=False, t_min=t_min, t_max=t_max,
                return_alphas=reuse)
            ray_indices, t_starts, t_ends = res[:3]
            acc_map, depth_map, _ = secondary_rendering(t_starts, t_ends, ray_indices=ray_indices, n_rays=n_rays,
                                                        alpha_fn=alpha_fn, chunk_size=self.secondary_shader_chunk,
                                                        alphas=res[3] if reuse else None)
        return 1.0 - acc_map, depth_map

    # ---- one ray batch (:224-443) ------------------------------------------------------------------------------
    def forward_(self, rays, relighting=False, stratified_u=None, curvature_dirs=None):
        if self.stage != 0 and self.emitter is None:
            raise RuntimeError("stage 1 needs model.light (envlight-mip-cube)")
        n_rays = rays.shape[0]
        rays_o, rays_d = rays[:, 0:3].contiguous(), rays[:, 3:6].contiguous()
        dev = rays.device

        # the reference evaluates the curvature term whenever it trains with FD normals (:251, :287); the
        # ``curvature`` key lets a config without lambda_curvature skip that extra field evaluation
        has_laplace = (self.geometry.grad_type == "finite_difference" and self.training
                       and bool(self.config.get("curvature", True)))

        def rgb_normal_alpha_fn(t_starts, t_ends, ray_indices):
            sdf, sdf_grad, normal, alphas, feature = self._stencil(rays_o, rays_d, ray_indices, t_starts,
                                                                   t_ends, True)
            t_dirs = rays_d[ray_indices]
            positions = self.geometry.last_points(ray_indices) if hasattr(self.geometry, "last_points") else None
            if positions is None:      # (the stencil kernels' own midpoints are these values bit for bit: geometry.sdf7_from_rays)
                positions = rays_o[ray_indices] + t_dirs * (t_starts + t_ends)[..., None] / 2.0
            colors = self.texture(feature, t_dirs, normal, positions, self.emitter, self.stage)
            if has_laplace:
                return colors, normal, alphas, sdf, sdf_grad, self.geometry.curvature(positions, sdf_grad,
                                                                                      curvature_dirs)
            return colors, normal, alphas, sdf, sdf_grad

        with torch.no_grad():
            ray_indices, t_starts, t_ends = self.occupancy_grid.sampling(
                rays_o, rays_d, alpha_fn=self._alpha_fn(rays_o, rays_d),
                render_step_size=self.render_step_size,
                stratified=self.randomized and stratified_u is None, stratified_u=stratified_u,
                cone_angle=0.0, alpha_thre=0.0)
        if self.after_sampling is not None:
            # Work that does not depend on the samples and that the caller wants issued HERE (rise_sdf_amd.step.TrainStep:
            # the environment prefilter, systems/split_occ.py:151-152).  The sampling call above ends in the step's one host
            # read; whatever is enqueued before it only delays that read, whereas the prefilter's ~2 ms of kernels enqueued
            # after it give the host that long a head start on issuing the render stage.
            hook, self.after_sampling = self.after_sampling, None
            hook()
        rgb_map, normal_map, acc_map, depth_map, extras = rendering_with_normals_sdf(
            t_starts, t_ends, ray_indices=ray_indices, n_rays=n_rays, rgb_alpha_fn=rgb_normal_alpha_fn,
            render_bkgd=None, has_laplace=has_laplace, color_dim=7 if self.stage == 0 else 24)

        diff_rgb_map, spec_rgb_map, blend_map = rgb_map[..., :3], rgb_map[..., 3:6], rgb_map[..., 6:7]
        if self.stage != 0:
            diff_rgb_pbr_map, spec_rgb_pbr_map = rgb_map[..., 7:10], rgb_map[..., 10:13]
            spec_ref_map, spec_light_map = rgb_map[..., 13:16], rgb_map[..., 16:19]
            albedo_map, metallic_map, roughness_map = rgb_map[..., 19:22], rgb_map[..., 22:23], rgb_map[..., 23:]
        if (self.masked_secondary and self.training and not relighting and self.config.get("indirect_pred", False)):
            # The same blend without the host read of ``torch.nonzero`` (models/split_mixed_occ.py:291): every ray goes
            # through the secondary pass, rays with opacity <= 0.5 with an empty marching interval, and the blend is a
            # select.  Per ray the arithmetic is the reference's; unselected rays contribute nothing, forward or backward.
            valid = acc_map > 0.5                                            # [N,1]
            sec_o = rays_o + depth_map * rays_d
            wo = -rays_d
            sec_d = 2 * torch.sum(wo * normal_map, dim=-1, keepdim=True) * normal_map - wo
            tr, sec_depth = self.compute_indirect_radiance(sec_o.detach().contiguous(), sec_d.detach().contiguous(),
                                                           valid=valid[:, 0])
            tr, sec_depth = tr.clamp(0, 1).detach(), sec_depth.detach()
            self._last_secondary = {"valid": valid, "sec_o": sec_o.detach(), "sec_d": sec_d.detach(), "tr": tr,
                                    "sec_depth": sec_depth}
            sec_feature = self.geometry(sec_o, with_grad=False, with_feature=True, input_grad=True)[1]
            sec_rgb = self.texture.secondary_shading(sec_feature, sec_d, normal_map)
            spec_rgb_map = torch.where(valid, tr * spec_rgb_map + (1 - tr) * sec_rgb, spec_rgb_map)
            if self.stage != 0:
                spec_rgb_pbr_map = torch.where(valid, tr * spec_rgb_pbr_map + (1 - tr) * sec_rgb, spec_rgb_pbr_map)
            valid_indices = None
        else:
            valid_indices = torch.nonzero(acc_map > 0.5)[..., 0]
        if valid_indices is not None and valid_indices.numel() > 0 and self.config.get("indirect_pred", False):
            sec_o = rays_o[valid_indices] + depth_map[valid_indices] * rays_d[valid_indices]
            wo = -rays_d[valid_indices]
            nv = normal_map[valid_indices]
            sec_d = 2 * torch.sum(wo * nv, dim=-1, keepdim=True) * nv - wo
            tr, sec_depth = self.compute_indirect_radiance(sec_o.detach().contiguous(), sec_d.detach().contiguous())
            tr, sec_depth = tr.clamp(0, 1).detach(), sec_depth.detach()
            # (references only, no copies: what a parity test needs to line the secondary pass up with the oracle)
            self._last_secondary = {"valid_indices": valid_indices, "sec_o": sec_o.detach(), "sec_d": sec_d.detach(),
                                    "tr": tr, "sec_depth": sec_depth}
            # sec_o carries a graph (depth_map -> weights): the feature query must propagate d/d(xyz) through the
            # encoding's xyz pass-through too (models/split_mixed_occ.py:315 goes through tcnn + nn.Linear input grads)
            sec_feature = self.geometry(sec_o, with_grad=False, with_feature=True, input_grad=True)[1]
            sec_rgb = self.texture.secondary_shading(sec_feature, sec_d, nv)
            spec_rgb_map = spec_rgb_map.clone()
            spec_rgb_map[valid_indices] = tr * spec_rgb_map[valid_indices] + (1 - tr) * sec_rgb
            if self.stage != 0 and not relighting:
                spec_rgb_pbr_map = spec_rgb_pbr_map.clone()
                spec_rgb_pbr_map[valid_indices] = tr * spec_rgb_pbr_map[valid_indices] + (1 - tr) * sec_rgb
            elif self.stage != 0:
                # relighting (:322-331): the learned secondary radiance belongs to the training light, so smooth
                # pixels shade the point their reflection hits with the NEW environment (third bounce)
                rmask = (roughness_map[valid_indices] <= self.config.relighting_threshold)[..., 0]
                third_o = sec_o[rmask] + sec_depth[rmask] * sec_d[rmask]
                _, third_grad, third_feature = self.geometry(third_o, with_grad=True, with_feature=True)
                third_normal = F.normalize(third_grad, p=2, dim=-1, eps=1e-6)
                third_rgb = self.texture.secondary_shading_pbr(third_feature, sec_d[rmask], third_normal, third_o,
                                                               self.emitter)
                light_valid = spec_light_map[valid_indices]
                light_valid[rmask] = tr[rmask] * light_valid[rmask] + (1 - tr[rmask]) * third_rgb
                spec_light_map = spec_light_map.clone()
                spec_light_map[valid_indices] = light_valid
                spec_rgb_pbr_map = spec_ref_map * spec_light_map
        rgb = diff_rgb_map + spec_rgb_map

        out = {"comp_rgb": rgb, "comp_diffuse_rgb": diff_rgb_map, "comp_spec_rgb": spec_rgb_map,
               "comp_blend": blend_map, "comp_normal": normal_map, "opacity": acc_map, "depth": depth_map,
               "rays_valid": acc_map > 0,
               "num_samples": torch.full((1,), len(t_starts), dtype=torch.int32, device=dev)}   # (a fill kernel: no host-to-device copy)
        if self.stage != 0:
            out.update({"comp_rgb_phys": diff_rgb_pbr_map + spec_rgb_pbr_map,
                        "comp_diffuse_rgb_phys": diff_rgb_pbr_map, "comp_spec_rgb_phys": spec_rgb_pbr_map,
                        "comp_albedo": albedo_map, "comp_metallic": metallic_map, "comp_roughness": roughness_map})
        if self.training:
            weights = extras["weights"]
            # the host already holds the sample count (the sampler sized its outputs with it): hand it on, so that
            # dynamic_ray_sampling (systems/split_occ.py:160 ``.item()``) needs no second read
            out["num_samples_host"] = int(t_starts.shape[0])
            out.update({"sdf_samples": extras["sdf"], "sdf_grad_samples": extras["sdf_grad"],
                        "weights": weights.view(-1), "ray_indices": ray_indices.view(-1)})
            if has_laplace:
                out["sdf_laplace_samples"] = extras["sdf_laplace"]
            if ray_indices.numel() > 0:
                orient = torch.sum(rays_d[ray_indices] * extras["normals"], dim=-1, keepdim=True).clamp(min=0)
                out["normals_orientation_loss_map"] = ops.accumulate_along_rays(
                    weights, orient, packed_info=extras["packed_info"])
            else:
                out["normals_orientation_loss_map"] = torch.zeros_like(rgb[..., :1])
        bg = self.background_color if self.background_color is not None else torch.ones(3, device=dev)
        out_bg = {"comp_rgb": bg[None, :].expand(*rgb.shape), "num_samples": torch.zeros_like(out["num_samples"]),
                  "rays_valid": torch.zeros_like(out["rays_valid"])}
        out_full = {"comp_rgb": T.compose_srgb(out["comp_rgb"], bg, out["opacity"]),
                    "num_samples": out["num_samples"] + out_bg["num_samples"],
                    "rays_valid": out["rays_valid"] | out_bg["rays_valid"]}
        if self.stage != 0:
            out_bg["comp_rgb_phys"] = out_bg["comp_rgb"]
            comp = lambda k: T.compose_srgb(out[k], bg, out["opacity"])       # one kernel each way (O1)
            out_full.update({"comp_rgb_phys": comp("comp_rgb_phys"), "comp_spec_rgb": comp("comp_spec_rgb"),
                             "comp_spec_rgb_phys": comp("comp_spec_rgb_phys")})
        return {**out, **{k + "_bg": v for k, v in out_bg.items()}, **{k + "_full": v for k, v in out_full.items()}}

    def forward(self, rays, relighting=False, **kw):
        if self.training:
            out = self.forward_(rays, relighting=relighting, **kw)
        else:
            out = chunk_batch(self.forward_, self.config.get("ray_chunk", 4096), False, rays, relighting,
                              streams=int(self.config.get("eval_streams", 2)))
        return {**out, "inv_s": self.variance.inv_s}

    def train(self, mode=True):
        self.randomized = mode and self.config.get("randomized", True)
        return super().train(mode=mode)

    def eval(self):
        self.randomized = False
        return super().eval()

    def regularizations(self, out):
        losses = {}
        losses.update(self.geometry.regularizations(out))
        losses.update(self.texture.regularizations(out))
        return losses


@torch.no_grad()
def relight(model, rays, emitter, reference=None, fg_mask=None):
    """systems/split_occ.py:405-420: render ``rays`` under another environment light.  The emitter is swapped, its
    mips rebuilt once, the model evaluated with ``relighting=True``, and (when a reference image is given) the
    prediction rescaled per channel by the median ratio over the foreground, as the reference's test step does.
    Everything stays on the device."""
    old = model.emitter
    model.emitter = emitter
    try:
        emitter.build_mips()
        out = model(rays, relighting=True)
    finally:
        model.emitter = old
    pred = out["comp_rgb_phys_full"]
    if reference is not None:
        m = fg_mask.bool() if fg_mask is not None else torch.ones(pred.shape[0], dtype=torch.bool, device=pred.device)
        ratio, _ = (reference[m] / pred[m].clamp(min=1e-6)).median(dim=0)
        pred = pred.clone()
        pred[m] = (ratio * pred[m]).clamp(min=0.0, max=1.0)
    return pred, out


def _registry():
    from .registry import models
    return models
