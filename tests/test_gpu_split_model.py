"""Assembled split-mixed-occ model vs the composed oracle (oracle/split_mixed_occ.py): secondary-ray occlusion (R1,
models/split_mixed_occ.py:179-222,291-318), the stage-1 model (:295-303,344-352,416-432) and relighting with the
third bounce (:320-331; systems/split_occ.py:405-420).  Radiance-like outputs at the north_star's 1e-4 relative."""
import pytest
import torch

import oracle
from oracle import split_mixed_occ as OS
from oracle import texture as OT
from helpers import camera_rays, rel_err, sphere_binary
from test_gpu_model import assert_grads_tight, hip_sdf7, oracle_params, split_config

pytestmark = pytest.mark.gpu
LIGHT = {"name": "envlight-mip-cube", "envlight_config": {"scale": 0.5, "bias": 0.25, "base_res": 64, "hdr_filepath": None}}


def _nets(tex, grad=True):
    out = {}
    for name in ("albedo", "metallic", "roughness", "env", "secondary"):
        net = getattr(tex, name + "_network")
        out[name] = [{"w": m.weight.detach().cpu().clone().requires_grad_(grad),
                      "b": m.bias.detach().cpu().clone().requires_grad_(grad)}
                     for m in net.layers if isinstance(m, torch.nn.Linear)]
    return out


def _build(dev, stage1, seed=0):
    import rise_sdf_amd as R
    torch.manual_seed(seed)
    cfg = split_config(indirect=True)
    cfg["curvature"] = False
    if stage1:
        cfg["split_sum_kick_in_step"] = 0
        cfg["relighting_threshold"] = 0.6
        cfg["light"] = LIGHT
    model = R.make("split-mixed-occ", cfg).to(dev)
    model.train()
    with torch.no_grad():
        model.geometry.encoding.encoding.encoding.params.mul_(1000.0)      # lumpy blob: some reflections are occluded
        l0 = model.geometry.network.layers[0]
        l0.weight_v[:, 3:] = torch.randn_like(l0.weight_v[:, 3:]) * 0.3
        model.variance.variance.fill_(0.6)            # sharp surface: opaque pixels, secondary rays fire
        if stage1:
            model.texture.FG_LUT = OT.synthetic_fg_lut(64).to(dev)
    model.occupancy_grid.binaries = sphere_binary(128, 0.2, 0.9).to(dev)[None]
    model.background_color = torch.ones(3, device=dev)
    model.update_step(0, 0)
    assert model.stage == (1 if stage1 else 0)
    return model


def _params(model):
    meta, table, mlp, var = oracle_params(model)
    P = dict(table=table, meta=meta, mlp=mlp, var=var, nets=_nets(model.texture),
             binary=model.occupancy_grid.binaries[0].cpu(), radius=1.5, fd_eps=model.geometry._finite_difference_eps,
             render_step_size=model.render_step_size, sec_near=model.secondary_near_plane,
             sec_far=model.secondary_far_plane, sec_steps=model.num_samples_per_secondary_ray,
             background=torch.ones(3), relighting_threshold=model.config.get("relighting_threshold", 0.0))
    if model.emitter is not None:
        P["emitter_base"] = model.emitter.base.detach().cpu().clone().requires_grad_(True)
        P["fg_lut"] = model.texture.FG_LUT.detach().cpu()
    return P


def _run_override(model):
    """The override dictionary of the last _run on ``model`` (sample sets, secondary rays, stencil and alpha values)."""
    return model._last_override


def _same_up_to_borderline(a, b, slack):
    key = lambda r, t: set(zip(r.tolist(), t.contiguous().view(torch.int32).tolist()))   # noqa: E731
    return len(key(a[0], a[1]) ^ key(b[0], b[1])) <= slack


def _n_different(a, b):
    key = lambda r, t: set(zip(r.tolist(), t.contiguous().view(torch.int32).tolist()))   # noqa: E731
    return len(key(a[0], a[1]) ^ key(b[0], b[1]))


def _all_param_grads(model, P):
    """name -> (HIP grad, oracle grad) for every network of the model."""
    hip, ref = {}, {}
    lin = [m for m in model.geometry.network.layers if isinstance(m, torch.nn.Linear)]
    for i, (m, p) in enumerate(zip(lin, P["mlp"])):
        for name, key in (("weight_v", "v"), ("weight_g", "g"), ("bias", "b")):
            if p[key].grad is not None:
                hip[f"sdf{i}.{name}"], ref[f"sdf{i}.{name}"] = getattr(m, name).grad, p[key].grad
    for net in ("albedo", "metallic", "roughness", "env", "secondary"):
        layers = [m for m in getattr(model.texture, net + "_network").layers if isinstance(m, torch.nn.Linear)]
        for i, (m, p) in enumerate(zip(layers, P["nets"][net])):
            if p["w"].grad is not None:
                hip[f"{net}{i}.w"], ref[f"{net}{i}.w"] = m.weight.grad, p["w"].grad
                hip[f"{net}{i}.b"], ref[f"{net}{i}.b"] = m.bias.grad, p["b"].grad
    if P["var"].grad is not None:
        hip["variance"], ref["variance"] = model.variance.variance.grad.reshape(1), P["var"].grad.reshape(1)
    return hip, ref


def _run(dev, model, rays, u, P, stage, relighting):
    """HIP forward, then the oracle on the HIP path's own sample sets / secondary rays (after checking that the
    oracle's own ones agree with them up to borderline-visibility samples)."""
    rd = rays.to(dev)
    with torch.no_grad():
        ro_d, rd_d = rd[:, :3].contiguous(), rd[:, 3:].contiguous()
        prim = model.occupancy_grid.sampling(ro_d, rd_d, alpha_fn=model._alpha_fn(ro_d, rd_d),
                                             render_step_size=model.render_step_size, stratified_u=u.to(dev),
                                             cone_angle=0.0, alpha_thre=0.0)
    out = model.forward_(rd, relighting=relighting, stratified_u=u.to(dev))
    last = model._last_secondary
    with torch.no_grad():
        so, sd = last["sec_o"].contiguous(), last["sec_d"].contiguous()
        step = (model.secondary_far_plane - model.secondary_near_plane) / (model.num_samples_per_secondary_ray - 1)
        sec = model.occupancy_grid.sampling(so, sd, alpha_fn=model._alpha_fn(so, sd),
                                            near_plane=model.secondary_near_plane, far_plane=model.secondary_far_plane,
                                            render_step_size=step, stratified=False, return_alphas=True)
        sec, sec_alphas = sec[:3], sec[3]
    cpu = lambda t3: tuple(t.cpu() for t in t3)   # noqa: E731
    ov = {"primary": cpu(prim), "sec_rays": (so.cpu(), sd.cpu()), "secondary": cpu(sec)}
    big = model.geometry.network.n_neurons > 64
    if big:
        # L = 16 / H = 128: the occlusion pass's alphas sit behind the same 1 / eps x inv_s amplifier as the primary ones and
        # have no stencil override of their own: handed in by VALUE (no gradient flows through that pass) after the
        # comparison with the oracle's own below
        ov["sec_alphas"] = sec_alphas.cpu()
    # the primary stencil VALUES of the HIP path (oracle.volume_sdf, sdf7_given): removes the 1/eps amplification of forward
    # ulps, so that the gradients below are held to SURVEY 8(d)'s 1e-4 / 1e-3 instead of a cosine
    ov["sdf7"] = hip_sdf7(model, rays, *ov["primary"])
    # ... and its alpha VALUES: these models run at inv_s = 403 (variance 0.6), where the reference's weight backward turns an
    # ulp of alpha into O(1) of d_alpha on saturated rays (tests/test_gpu_late_regime.py); C1 itself is bit-exact on equal alphas
    with torch.no_grad():
        ov["alphas"] = model._alpha_fn(ro_d, rd_d)(prim[1], prim[2], prim[0]).cpu()
    model._last_override = ov                              # (tests/test_gpu_big_model.py re-renders the oracle on these)
    ref = OS.render(rays, P, stage=stage, indirect=True, relighting=relighting, stratified_u=u, override=ov)
    # ... and the stencil itself agrees to fp32 rounding (relative to max |sdf|; the H = 128 chain sits at 7e-7 of its largest
    # ACTIVATION, as in tests/test_gpu_late_regime.py)
    assert rel_err(ov["sdf7"], ref["sdf7"]) < (3e-6 if model.geometry.network.n_neurons <= 64 else 8e-6)
    # ... and so do the alphas that were handed in (an override that is never compared hides a regression of
    # the HIP alpha kernel).  With sdf7_given the oracle's OWN alphas are its get_alpha on the same stencil values: what
    # remains is the fp32 rounding of normalize / sigmoid, amplified by inv_s (403 here) in the sigmoid's argument
    a_err = (ov["alphas"] - ref["alphas_own"]).abs()
    assert float(a_err.max()) < 2e-5 and float(a_err.mean()) < 1e-6, (float(a_err.max()), float(a_err.mean()))
    # the oracle's OWN sampling / secondary rays agree with the HIP path's up to borderline samples and fp32 depth;
    # the number of borderline samples that actually differed is reported (no silent slack)
    d_prim = _n_different(ref["own_primary"], ov["primary"])
    d_sec = _n_different(ref["own_secondary"], ov["secondary"])
    print("borderline-visibility samples that differ: primary %d of %d, secondary %d of %d" %
          (d_prim, ov["primary"][0].numel(), d_sec, ov["secondary"][0].numel()))
    assert _same_up_to_borderline(ref["own_primary"], ov["primary"], max(3, ov["primary"][0].numel() // 2000))
    assert torch.equal(ref["valid_indices"], last["valid_indices"].cpu())
    assert torch.allclose(ref["own_sec_rays"][0], so.cpu(), rtol=1e-4, atol=2e-5)
    assert torch.allclose(ref["own_sec_rays"][1], sd.cpu(), rtol=1e-3, atol=1e-3)
    assert _same_up_to_borderline(ref["own_secondary"], ov["secondary"], max(3, ov["secondary"][0].numel() // 500))
    if big:
        sa_err = (ov["sec_alphas"] - ref["sec_alphas_own"]).abs()
        print("occlusion-pass alphas, HIP vs the oracle's own: max %.2e, mean %.2e, above 1e-3: %d of %d" %
              (float(sa_err.max()), float(sa_err.mean()), int((sa_err > 1e-3).sum()), sa_err.numel()))
        assert float(sa_err.mean()) < 1e-4 and int((sa_err > 1e-2).sum()) <= max(2, sa_err.numel() // 200)
    assert torch.allclose(last["tr"].cpu(), ref["tr"], rtol=1e-4, atol=2e-5), float((last["tr"].cpu() - ref["tr"]).abs().max())
    assert float(ref["tr"].min()) < 0.5 < float(ref["tr"].max()), "scene must have occluded and unoccluded reflections"
    return out, ref


def test_secondary_rays_stage0_vs_oracle(dev):
    model = _build(dev, stage1=False)
    rays = camera_rays(20, 20, seed=2)
    u = torch.rand(rays.shape[0], generator=torch.Generator().manual_seed(3))
    P = _params(model)
    out, ref = _run(dev, model, rays, u, P, 0, False)
    assert int(ref["valid_indices"].numel()) > 40
    for k in ("comp_rgb", "comp_spec_rgb", "comp_diffuse_rgb", "comp_blend", "opacity", "comp_rgb_full"):
        assert torch.allclose(out[k].cpu(), ref[k], rtol=1e-4, atol=2e-5), k
    g = torch.randn(ref["comp_rgb_full"].shape, generator=torch.Generator().manual_seed(4))
    (ref["comp_rgb_full"] * g).sum().backward()
    (out["comp_rgb_full"] * g.to(dev)).sum().backward()
    gt = model.geometry.encoding.encoding.encoding.params.grad.cpu()
    assert torch.nn.functional.cosine_similarity(gt[None], P["table"].grad[None]).item() > 0.99999
    # every parameter gradient at SURVEY 8(d)'s tolerances (the oracle ran on the HIP path's stencil values, _run)
    hip, ref_g = _all_param_grads(model, P)
    assert len(ref_g) >= 3 * 3 + 2 * 5 and "secondary0.w" in ref_g
    assert_grads_tight(hip, ref_g, gt, P["table"].grad)


def test_stage1_model_vs_oracle(dev):
    model = _build(dev, stage1=True)
    rays = camera_rays(16, 16, seed=2)
    u = torch.rand(rays.shape[0], generator=torch.Generator().manual_seed(3))
    P = _params(model)
    model.emitter.build_mips()
    out, ref = _run(dev, model, rays, u, P, 1, False)
    for k in ("comp_rgb", "comp_rgb_phys", "comp_diffuse_rgb_phys", "comp_spec_rgb_phys", "comp_spec_rgb", "comp_albedo",
              "comp_metallic", "comp_roughness", "comp_rgb_full", "comp_rgb_phys_full", "comp_spec_rgb_full",
              "comp_spec_rgb_phys_full"):
        assert torch.allclose(out[k].cpu(), ref[k], rtol=1e-4, atol=2e-5), k
    g = torch.randn(ref["comp_rgb_phys_full"].shape, generator=torch.Generator().manual_seed(5))
    (ref["comp_rgb_phys_full"] * g).sum().backward()
    (out["comp_rgb_phys_full"] * g.to(dev)).sum().backward()
    assert rel_err(model.emitter.base.grad, P["emitter_base"].grad) < 1e-3
    hip, ref_g = _all_param_grads(model, P)
    gt = model.geometry.encoding.encoding.encoding.params.grad.cpu()
    # stage 1 runs through the prefiltered environment (the oracle's prefilters and cube lookups are fp64 dense-weight
    # restatements, oracle/envlight.py) and the FG-LUT: every parameter gradient within 2e-3 of its tensor's largest entry
    # (measured: up to 1.1e-3 on the radiance networks; the old gate looked at ONE layer at 2e-3 and the table by cosine)
    assert_grads_tight(hip, ref_g, gt, P["table"].grad, mlp_tol=2e-3, table_tol=2e-3)


def test_relight_third_bounce_vs_oracle(dev):
    """The relit render itself (not only its shape): swap the emitter, rebuild its mips, third-bounce shading of smooth
    pixels, median-ratio rescale against a reference image (systems/split_occ.py:405-420)."""
    import rise_sdf_amd as R
    from rise_sdf_amd.split_mixed_occ import relight
    model = _build(dev, stage1=True)
    model.eval()
    model.config["ray_chunk"] = 4096
    rays = camera_rays(16, 16, seed=2)
    new_light = R.make("envlight-mip-cube", R.Config(LIGHT)).to(dev)
    with torch.no_grad():
        new_light.base.copy_(torch.rand(new_light.base.shape, generator=torch.Generator().manual_seed(9)).to(dev) * 2.0)
    ref_img = torch.rand(rays.shape[0], 3, generator=torch.Generator().manual_seed(10))
    pred, out = relight(model, rays.to(dev), new_light, reference=ref_img.to(dev), fg_mask=None)
    P = _params(model)
    P["emitter_base"] = new_light.base.detach().cpu()
    # eval mode: no jitter; the HIP path's sample sets, as in _run
    model.emitter = new_light
    try:
        model.train()
        model.randomized = False
        with torch.no_grad():
            new_light.build_mips()
            _, ref = _run(dev, model, rays, torch.zeros(rays.shape[0]), P, 1, True)
    finally:
        model.eval()
    assert int(ref["rmask"].sum()) > 0, "some pixels must take the third bounce"
    with torch.no_grad():
        for k in ("comp_rgb_phys", "comp_spec_rgb_phys", "comp_rgb_phys_full"):
            assert torch.allclose(out[k].cpu(), ref[k], rtol=1e-4, atol=2e-5), k
        p = ref["comp_rgb_phys_full"]
        ratio, _ = (ref_img / p.clamp(min=1e-6)).median(dim=0)
        want = (ratio * p).clamp(0.0, 1.0)
        assert torch.allclose(pred.cpu(), want, rtol=2e-4, atol=5e-5)


def test_round3_fault_order_with_the_normal_fold():
    """VERDICT r03 item 2.  In round 3 a first version of the normal-map fold (the [S,3] normals accumulated inside the opacity /
    depth pass) aborted with a GPU memory fault in test_stage1_model_vs_oracle -- only when that test ran after
    tests/test_gpu_ops.py (gpurun_out/r03g/repro.log) -- and was withdrawn unexplained.  The fold was re-implemented in round 4
    (rsdf_opacity_depth_normal_fwd / _bwd) and is the default; this test runs exactly that order, in a process of its own, with
    RSDF_CHECK=1: every packed_info handed to a per-ray kernel is validated against the sample arrays it is used with (the
    suspects: a stale cached pack, under-sized capacity buffers)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RSDF_CHECK="1", RSDF_FOLD_NORMALS="1", PYTHONFAULTHANDLER="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(root, "tests", "test_gpu_ops.py"),
                        os.path.join(root, "tests", "test_gpu_split_model.py") + "::test_stage1_model_vs_oracle"],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=root)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    assert "passed" in r.stdout and "failed" not in r.stdout
