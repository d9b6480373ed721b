"""Oracle for the ASSEMBLED split-mixed-occ model (stage 0 / stage 1, relighting): a CPU
restatement of models/split_mixed_occ.py:179-222 (compute_indirect_radiance), :224-443 (forward_) and :320-331 (third
bounce) that composes the pieces the other oracle modules restate -- marcher + visibility pruning
(oracle.ray_marching), field / FD normals / NeuS alpha (oracle.neus_geometry_render), radiance branch
(oracle.texture), environment light (oracle.envlight) and compositing.  TEST INFRASTRUCTURE ONLY.

The surviving sample sets depend on fp32 alphas through T >= 1e-4 and the secondary rays start at the composited depth,
so two correct fp32 implementations can differ in a few borderline samples.  ``override`` lets a test hand in the sample
sets / secondary rays of the implementation under test after checking that they agree with the oracle's own up to such
borderline entries; everything downstream of them is then compared value for value.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import (accumulate_along_rays, neus_geometry_render, ray_marching, render_weight_from_alpha, volume_sdf)
from . import envlight as E
from . import texture as T

ROI = lambda r: torch.tensor([-r, -r, -r, r, r, r], dtype=torch.float32)   # noqa: E731


def emitter_fns(base):
    """lib/pbr/light.py:169-206 for a base cube map [6,R,R,3]: build_mips once -> (eval_diffuse, eval_specular)."""
    spec, diffuse = E.build_mips(base.double())            # fp64 prefilters (dense weights), as the envlight tests

    def eval_diffuse(n):
        return E.cube_sample_linear(diffuse, n.double()).to(n.dtype)

    def eval_specular(d, roughness):
        return E.cube_sample_mip(spec, d.double(), E.get_mip(roughness.double(), len(spec))[..., 0]).to(d.dtype)
    return eval_diffuse, eval_specular


def render(rays, P, *, stage, indirect, relighting=False, stratified_u=None, override=None):
    """P: dict(table, meta, mlp, var, nets{albedo,roughness,metallic,env,secondary}, binary [128^3 bool], radius,
    fd_eps, render_step_size, sec_near, sec_far, sec_steps, background [3], fg_lut, emitter_base (stage 1),
    relighting_threshold).  Returns the output dictionary of models/split_mixed_occ.py:340-443 (the keys the systems
    consume) plus the intermediates a parity test needs."""
    override = override or {}
    r = P["radius"]
    roi = ROI(r)
    n_rays = rays.shape[0]
    ro, rd = rays[:, :3].contiguous(), rays[:, 3:].contiguous()
    field = dict(radius=r, fd_eps=P["fd_eps"])

    def alpha_fn_for(o, d):
        rr = torch.cat([o, d], -1)

        def alpha_fn(ts, te, ri):
            with torch.no_grad():
                return neus_geometry_render(rr, ri, ts, te, P["table"], P["meta"], P["mlp"], P["var"], **field)["alphas"]
        return alpha_fn

    # ---- primary rays: visibility-pruned sampling (:264-272), field, radiance, compositing (:274-289) -------------
    own_primary = ray_marching(ro, rd, scene_aabb=roi, grid_roi=roi, grid_binary=P["binary"], near_plane=0.0,
                               far_plane=1e10, render_step_size=P["render_step_size"], stratified_u=stratified_u,
                               alpha_fn=alpha_fn_for(ro, rd))
    ri, ts, te = override.get("primary", own_primary)
    # override["sdf7"] [S,7]: the stencil VALUES of the implementation under test (oracle.volume_sdf, sdf7_given)
    # override["alphas"] [S]: the alpha VALUES of the implementation under test (neus_geometry_render, alphas_given): in a sharp
    # field the reference's weight backward amplifies an ulp of alpha (render_weight.cu:139-151)
    ref = neus_geometry_render(rays, ri, ts, te, P["table"], P["meta"], P["mlp"], P["var"], **field,
                               sdf7_given=override.get("sdf7"), alphas_given=override.get("alphas"))
    pos = ro[ri] + rd[ri] * ((ts + te) / 2.0)[:, None]
    if stage == 0:
        colors = T.texture_stage0(ref["feature"], rd[ri], ref["normal"], pos, P["nets"])
    else:
        ev_d, ev_s = emitter_fns(P["emitter_base"])
        colors = T.texture_stage1(ref["feature"], rd[ri], ref["normal"], pos, P["nets"], P["fg_lut"], ev_d, ev_s)
    comp = accumulate_along_rays(ref["weights"], colors, ray_indices=ri, n_rays=n_rays)
    normal_map = ref["comp_normal"]
    acc, depth = ref["opacity"], ref["depth"]
    diff, spec, blend = comp[:, :3], comp[:, 3:6], comp[:, 6:7]
    out = {"own_primary": own_primary, "primary": (ri, ts, te), "sdf7": ref["sdf7"], "alphas_own": ref["alphas_own"]}
    if stage != 0:
        diff_pbr, spec_pbr = comp[:, 7:10], comp[:, 10:13]
        spec_ref, spec_light = comp[:, 13:16], comp[:, 16:19]
        albedo, metallic, roughness = comp[:, 19:22], comp[:, 22:23], comp[:, 23:]

    # ---- secondary (reflection) rays (:291-332) -----------------------------------------------------------------------
    valid = torch.nonzero(acc[:, 0] > 0.5)[:, 0]
    out["valid_indices"] = valid
    if valid.numel() > 0 and indirect:
        sec_o = ro[valid] + depth[valid] * rd[valid]
        wo = -rd[valid]
        nv = normal_map[valid]
        sec_d = 2 * torch.sum(wo * nv, -1, keepdim=True) * nv - wo
        out["own_sec_rays"] = (sec_o.detach(), sec_d.detach())
        so, sd = override.get("sec_rays", (sec_o.detach(), sec_d.detach()))
        step = (P["sec_far"] - P["sec_near"]) / (P["sec_steps"] - 1)
        a_fn = alpha_fn_for(so, sd)
        own_secondary = ray_marching(so, sd, scene_aabb=roi, grid_roi=roi, grid_binary=P["binary"],
                                     near_plane=P["sec_near"], far_plane=P["sec_far"], render_step_size=step,
                                     alpha_fn=a_fn)
        sri, sts, ste = override.get("secondary", own_secondary)
        out["own_secondary"], out["secondary"] = own_secondary, (sri, sts, ste)
        with torch.no_grad():
            sal = a_fn(sts, ste, sri) if sri.numel() else torch.zeros(0)
            out["sec_alphas_own"] = sal
            # override["sec_alphas"]: the occlusion pass's alpha VALUES of the implementation under test (no gradient flows
            # through this pass): at L = 16 the FD normal's 1 / eps and inv_s turn fp32 ulps of the stencil into 1e-3 of an
            # alpha next to a surface, and tr = 1 - sum w is compared at 1e-4 downstream
            sal = override.get("sec_alphas", sal)
            sw, _ = render_weight_from_alpha(sal, ray_indices=sri, n_rays=valid.numel())
            sacc = accumulate_along_rays(sw, None, ray_indices=sri, n_rays=valid.numel())
            sdepth = accumulate_along_rays(sw, ((sts + ste) / 2.0)[:, None], ray_indices=sri, n_rays=valid.numel())
            tr = (1.0 - sacc).clamp(0, 1)
        out["tr"], out["sec_depth"] = tr, sdepth
        # feature at the hit point WITH its graph (sec_o depends on depth -> weights; the reference goes through tcnn's
        # input gradient and nn.Linear, :315): the torch-op hash grid of oracle/analytic.py differentiates in x
        from . import analytic as OA
        sec_feature = OA.field(sec_o, P["table"], P["meta"], P["mlp"], r)
        sec_in = torch.cat([sec_feature, T.sh_encode((sec_d + 1.0) / 2.0, 5), nv], -1)
        sec_rgb = torch.sigmoid(T.relu_mlp(sec_in, P["nets"]["secondary"]))
        spec = spec.clone()
        spec[valid] = tr * spec[valid] + (1 - tr) * sec_rgb
        if stage != 0 and not relighting:
            spec_pbr = spec_pbr.clone()
            spec_pbr[valid] = tr * spec_pbr[valid] + (1 - tr) * sec_rgb
        elif stage != 0:
            rmask = (roughness[valid] <= P["relighting_threshold"])[:, 0]
            third_o = sec_o[rmask] + sdepth[rmask] * sec_d[rmask]
            _, third_grad, third_feature = volume_sdf(third_o, P["table"], P["meta"], P["mlp"], radius=r, fd_eps=P["fd_eps"])
            third_n = F.normalize(third_grad, p=2, dim=-1, eps=1e-6)
            third_rgb = T.secondary_shading_pbr(third_feature, sec_d[rmask], third_n, third_o, P["nets"], P["fg_lut"],
                                                ev_d, ev_s)
            light_valid = spec_light[valid].clone()
            light_valid[rmask] = tr[rmask] * light_valid[rmask] + (1 - tr[rmask]) * third_rgb
            spec_light = spec_light.clone()
            spec_light[valid] = light_valid
            spec_pbr = spec_ref * spec_light
            out["third_rgb"], out["rmask"] = third_rgb, rmask

    # ---- compose (:334-443) ----------------------------------------------------------------------------------------------------
    rgb = diff + spec
    bg = P["background"][None, :]
    full = lambda c: T.rgb_to_srgb(c + bg * (1.0 - acc)).clamp(0, 1)   # noqa: E731
    out.update({"comp_rgb": rgb, "comp_diffuse_rgb": diff, "comp_spec_rgb": spec, "comp_blend": blend,
                "comp_normal": normal_map, "opacity": acc, "depth": depth, "comp_rgb_full": full(rgb),
                "weights": ref["weights"], "sdf": ref["sdf"], "sdf_grad": ref["sdf_grad"]})
    if stage != 0:
        out.update({"comp_rgb_phys": diff_pbr + spec_pbr, "comp_diffuse_rgb_phys": diff_pbr,
                    "comp_spec_rgb_phys": spec_pbr, "comp_albedo": albedo, "comp_metallic": metallic,
                    "comp_roughness": roughness, "comp_rgb_phys_full": full(diff_pbr + spec_pbr),
                    "comp_spec_rgb_full": full(spec), "comp_spec_rgb_phys_full": full(spec_pbr)})
    return out
