"""The default fp32 SDF field (the two-part fp16 "x2" kernels, csrc/mlp_x2.hip) where a 40k-step run ends up, and its guards.

The reference's SDF network is plain fp32 with fp32's range (models/network_utils.py:109-157); the x2 form has a forward
range (|input| < 255, |weight| < 1023, |activation| < 454) and a backward dynamic range (one power-of-two scale per launch
for the gradient images).  Tests here:
  * the c1 model at L = 16 / T = 2^19, H = 64 and 128, against the ORACLE at variance 0.6 and 0.75 (inv_s 403 and 1808), a
    field sharp enough that alpha saturates to exactly 1, jitter on -- with RSDF_X2 unset (the shipped path);
  * one launch with a single d_sdf row at 2^30 times the typical magnitude: every other row's d(hash features) must equal the
    range-free kernels' at 1e-4 (the range guard reroutes the launch on the device);
  * the guard does NOT reroute an ordinary launch, and a forced reroute equals RSDF_X2=0;
  * a forward range violation produces the named error at the next host read instead of NaN losses."""
import ctypes
import sys
import os

import pytest
import torch

import oracle
from helpers import camera_rays, rel_err
from test_gpu_x2 import _field_inputs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from rise_sdf_amd import ops as o
    return o


@pytest.mark.parametrize("variance", [0.6, 0.75])
@pytest.mark.parametrize("hidden", [64, 128])
def test_c1_late_training_regime_vs_oracle(dev, hidden, variance, monkeypatch):
    """bench.py's c1 model (16 levels, 2^19 entries, base 32, fused x2 kernels) at the END of a training run instead of its
    start: inv_s = exp(10 variance) = 403 / 1808 (the reference's VarianceNetwork, models/split_mixed_occ.py:21-56, learns it
    upwards from 20), a field with a zero crossing, so that alpha = clip((p - n + 1e-5) / (p + 1e-5)) saturates to exactly 1
    at the surface and the weight backward's 1 / max(1 - alpha, 1e-10) (lib/nerfacc/cuda/csrc/render_weight.cu:139-151) is
    exercised; stratified jitter on; sampled the way the reference samples at that stage -- OccGridEstimator.sampling with
    alpha_fn, i.e. pruned at T < 1e-4 (models/split_mixed_occ.py:264-272).  RSDF_X2 unset: the shipped kernels.

    Each stage is compared with the oracle on the inputs the next stage amplifies: the oracle's field takes the HIP stencil
    values (1 / eps, then inv_s), its compositing takes the HIP alphas (the reference's weight backward turns one ulp of alpha
    into O(1) of d_alpha: measured here, its |d_alpha| reaches 1e4 where the closed form is bounded by 10) -- and both are
    first checked against the oracle's own values.  Gates: SURVEY 8(d)'s."""
    import rise_sdf_amd as R
    from rise_sdf_amd import _lib, fused
    monkeypatch.delenv("RSDF_X2", raising=False)
    monkeypatch.delenv("RSDF_X2_REROUTE", raising=False)
    sys.path.insert(0, ROOT)
    import bench
    torch.manual_seed(0)
    cfg = bench.c1_config(hidden=hidden)
    cfg["num_samples_per_ray"] = 256
    cfg["prune_by_visibility"] = True
    model = R.make("neus", cfg).to(dev)
    enc = model.geometry.encoding.encoding.encoding
    gen = torch.Generator().manual_seed(0)
    with torch.no_grad():
        enc.params.copy_(((torch.rand(enc.params.numel(), generator=gen) * 2 - 1) * 3e-2).to(dev))
        l0 = model.geometry.network.layers[0]
        l0.weight_v[:, 3:] = (torch.randn(l0.weight_v[:, 3:].shape, generator=gen) * 0.3).to(dev)
        model.variance.variance.fill_(variance)
        # (with the hash columns un-zeroed, weight_norm flattens the sphere init: the field is negative in the whole box;
        # shifted so that the central rays cross sdf = 0)
        model.geometry.network.layers[-1].bias[0] += 0.25
    model.train()
    model.geometry.update_step(0, 0)
    model.cos_anneal_ratio = 1.0
    assert model._fused_ok() and fused.x2_parts(35, hidden, 48, "fp32") == 2
    rays = camera_rays(48, 48, seed=21)
    n_rays = rays.shape[0]
    u = torch.rand(n_rays, generator=torch.Generator().manual_seed(22))
    rays_o, rays_d = rays[:, :3].contiguous().to(dev), rays[:, 3:].contiguous().to(dev)
    from test_gpu_model import assert_grads_tight, hip_sdf7, oracle_params
    eps = model.geometry._finite_difference_eps

    # ---- sampling: candidates bit-exact vs the oracle's marcher; the visibility filter on the HIP candidates' alphas must keep
    # exactly what the reference's render_visibility (sequential T >= 1e-4) keeps
    roi = torch.tensor([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5])
    ci, cs, ce = oracle.ray_marching(rays[:, :3].contiguous(), rays[:, 3:].contiguous(), scene_aabb=roi, near_plane=0.0,
                                     far_plane=1e10, render_step_size=model.render_step_size, stratified_u=u)
    alpha_fn = model._alpha_fn(rays_o, rays_d)
    with torch.no_grad():
        cand_alpha = alpha_fn(cs.to(dev), ce.to(dev), ci.to(dev)).cpu()
        ri_d, ts_d, te_d = model.occupancy_grid.sampling(rays_o, rays_d, alpha_fn=alpha_fn, render_step_size=model.render_step_size,
                                                         stratified_u=u.to(dev), cone_angle=0.0, alpha_thre=0.0)
    keep = oracle.render_visibility(cand_alpha, ray_indices=ci, n_rays=n_rays, early_stop_eps=1e-4)
    ri, ts, te = ci[keep], cs[keep], ce[keep]
    assert torch.equal(ri_d.cpu(), ri) and torch.equal(ts_d.cpu(), ts) and torch.equal(te_d.cpu(), te)
    assert 2000 < ri.numel() < ci.numel() // 4, (ri.numel(), ci.numel())

    # ---- forward
    out = model.render_samples(rays_o, rays_d, ri_d, ts_d, te_d, n_rays)
    sdf7 = hip_sdf7(model, rays, ri, ts, te)
    with torch.no_grad():
        alphas_hip = alpha_fn(ts_d, te_d, ri_d).cpu()
    meta2, table2, mlp2, var2 = oracle_params(model)
    if variance == 0.6:              # (the stencil values do not depend on the variance: once per width)
        with torch.no_grad():
            ref_free = oracle.neus_geometry_render(rays, ri, ts, te, table2, meta2, mlp2, var2, radius=1.5, fd_eps=eps)
        # (relative to max |sdf| = 0.18 here, six times smaller than in test_gpu_regimes.py's unshifted field: the H = 128 chain
        # sits at 7e-7 of its largest ACTIVATION)
        assert rel_err(sdf7, ref_free["sdf7"]) < (3e-6 if hidden == 64 else 8e-6)
        assert rel_err(out["sdf_samples"], ref_free["sdf"]) < 1e-5
    ref = oracle.neus_geometry_render(rays, ri, ts, te, table2, meta2, mlp2, var2, radius=1.5, fd_eps=eps, sdf7_given=sdf7,
                                      alphas_given=alphas_hip)
    # the regime itself: a zero crossing, saturated samples
    n_sat = int((alphas_hip == 1.0).sum())
    print(f"H = {hidden}, inv_s = {float(torch.exp(torch.tensor(10.0 * variance))):.0f}: {ri.numel()} of {ci.numel()} candidates "
          f"kept, {n_sat} alphas exactly 1, {int((alphas_hip > 0.999).sum())} above 0.999")
    # (inv_s 1808: alphas reach exactly 1 at every crossing; inv_s 403 with this step: they come within 1e-3 of it, and whether
    # some round to exactly 1 depends on the network's width)
    assert (n_sat > 100 if variance > 0.7 else int((alphas_hip > 0.999).sum()) > 100) \
        and float(ref["sdf"].min()) < -0.005 and float(ref["sdf"].max()) > 0.005, \
        "no zero crossing with saturated alphas: not the late-training regime"
    # the HIP alpha kernel against torch's chain on the same stencil values (get_alpha, models/split_mixed_occ.py:151-177):
    # sigmoid arguments reach +-100, one ulp of a CDF near 1 is 6e-8 absolute
    assert float((alphas_hip - ref["alphas_own"]).abs().max()) < 2e-6
    for k in ("opacity", "depth"):
        assert torch.allclose(out[k].cpu(), ref[k].detach(), rtol=1e-5, atol=1e-6), (k, float((out[k].cpu() - ref[k].detach()).abs().max()))
    assert float((out["sdf_grad_samples"].cpu() - ref["sdf_grad"].detach()).abs().max()) < 1e-4

    # ---- backward
    g = torch.Generator().manual_seed(23)
    go, gd = torch.randn(ref["opacity"].shape, generator=g), torch.randn(ref["depth"].shape, generator=g)
    ref["sdf"].retain_grad()
    ref["alphas"].retain_grad()
    ((ref["opacity"] * go).sum() + (ref["depth"] * gd).sum()).backward()
    ((out["opacity"] * go.to(dev)).sum() + (out["depth"] * gd.to(dev)).sum()).backward()
    torch.cuda.synchronize()
    gt = enc.params.grad
    assert bool(torch.isfinite(gt).all())
    # not vacuous: the centre rows' incoming gradient spans many binades below a maximum of order 10-100
    nz = ref["sdf"].grad[ref["sdf"].grad != 0].abs()
    print(f"  |d_alpha| max {float(ref['alphas'].grad.abs().max()):.3g}; centre-row d_sdf: {nz.numel()} non-zero, max "
          f"{float(nz.max()):.3g}, median {float(nz.median()):.3g}, min {float(nz.min()):.3g}; max |d_table| "
          f"{float(table2.grad.abs().max()):.3g}")
    assert nz.numel() > 1000 and float(nz.max()) > 1.0 and float(nz.max() / nz.min()) > 1e8
    assert float(table2.grad.abs().max()) > 1.0
    lin = [m for m in model.geometry.network.layers if isinstance(m, torch.nn.Linear)]
    hip_named, ref_named = {}, {}
    for i, (m, p) in enumerate(zip(lin, mlp2)):
        for name, key in (("weight_v", "v"), ("weight_g", "g"), ("bias", "b")):
            hip_named[f"{i}.{name}"], ref_named[f"{i}.{name}"] = getattr(m, name).grad, p[key].grad
    hip_named["variance"], ref_named["variance"] = model.variance.variance.grad.reshape(1), var2.grad.reshape(1)
    table_ref = table2.grad
    errs = {k: rel_err(hip_named[k], ref_named[k]) for k in ref_named}
    errs["table"] = float((gt.cpu() - table_ref).abs().max() / table_ref.abs().max())
    print("  HIP vs oracle: " + ", ".join(f"{k} {v:.1e}" for k, v in errs.items()))
    # ---- the gate.  SURVEY 8(d): 1e-4 on MLP parameters (3e-4 here: two fp32 summation orders of 1e4-1e5 terms,
    # test_gpu_model.assert_grads_tight) and 1e-3 on table rows -- UNLESS the reference's own backward is worse conditioned
    # than that in this case, which is MEASURED, not derived: the oracle's backward is taken again with every
    # stencil value moved to its fp32 neighbour (random signs, 3 trials).  Where the weight backward's residue amplifier is
    # active (render_weight.cu:139-151: |d_alpha| 6e3 .. 1.5e4 at inv_s 1808) its gradients move by 5e-4 .. 4e-3 of each
    # tensor's largest entry; at inv_s 403 they move by <= 1e-5 and SURVEY's gates stand as they are.  An implementation
    # cannot agree with the oracle more closely than the oracle agrees with itself one ulp away: the gate is 3 x that.
    from helpers import oracle_gradient_sensitivity
    leaves = {"table": table2, "variance": var2}
    for i, p in enumerate(mlp2):
        for name, key in (("weight_v", "v"), ("weight_g", "g"), ("bias", "b")):
            leaves[f"{i}.{name}"] = p[key]
    _, moved = oracle_gradient_sensitivity(
        lambda s7: oracle.neus_geometry_render(rays, ri, ts, te, table2, meta2, mlp2, var2, radius=1.5, fd_eps=eps,
                                               sdf7_given=s7, alphas_given=alphas_hip),
        leaves, sdf7, {"opacity": go, "depth": gd}, trials=3, seed=24)
    print("  oracle vs itself one ulp away: " + ", ".join(f"{k} {v:.1e}" for k, v in moved.items()))
    gates = {k: max(1e-3 if k == "table" else 3e-4, 3.0 * moved[k]) for k in errs}
    assert all(errs[k] < gates[k] for k in errs), {k: (errs[k], gates[k]) for k in errs if errs[k] >= gates[k]}
    if float(ref["alphas"].grad.abs().max()) < 100.0:
        assert max(errs.values()) < 1e-4 and max(moved.values()) < 1e-4, (errs, moved)
    else:
        # the regime this test is about: the reference's own backward is conditioned worse than SURVEY's gate
        assert max(moved.values()) > 3e-4, moved
    # ---- the range-free round-3 kernels (RSDF_X2=0: three bf16 parts, fp32's exponent range) on the same samples sit at the
    # same level: what is left is the reference's conditioning, not the x2 number format (was tools/debug/late_modes.py)
    monkeypatch.setenv("RSDF_X2", "0")
    for prm in model.parameters():
        prm.grad = None
    out3 = model.render_samples(rays_o, rays_d, ri_d, ts_d, te_d, n_rays)
    ((out3["opacity"] * go.to(dev)).sum() + (out3["depth"] * gd.to(dev)).sum()).backward()
    torch.cuda.synchronize()
    sdf7_3 = hip_sdf7(model, rays, ri, ts, te)
    with torch.no_grad():
        alphas_3 = alpha_fn(ts_d, te_d, ri_d).cpu()
    monkeypatch.delenv("RSDF_X2")
    assert rel_err(sdf7_3, sdf7) < 1e-5 and not torch.equal(sdf7_3, sdf7)       # another kernel family ran
    meta3, table3, mlp3, var3 = oracle_params(model)
    ref3 = oracle.neus_geometry_render(rays, ri, ts, te, table3, meta3, mlp3, var3, radius=1.5, fd_eps=eps, sdf7_given=sdf7_3,
                                       alphas_given=alphas_3)
    ((ref3["opacity"] * go).sum() + (ref3["depth"] * gd).sum()).backward()
    errs3 = {}
    for i, (m, p) in enumerate(zip(lin, mlp3)):
        for name, key in (("weight_v", "v"), ("weight_g", "g"), ("bias", "b")):
            errs3[f"{i}.{name}"] = rel_err(getattr(m, name).grad, p[key].grad)
    errs3["variance"] = rel_err(model.variance.variance.grad.reshape(1), var3.grad.reshape(1))
    errs3["table"] = float((enc.params.grad.cpu() - table3.grad).abs().max() / table3.grad.abs().max())
    print("  round-3 kernels vs oracle: " + ", ".join(f"{k} {v:.1e}" for k, v in errs3.items()))
    assert all(errs3[k] < gates[k] for k in errs3), {k: (errs3[k], gates[k]) for k in errs3 if errs3[k] >= gates[k]}
    st = _lib.poll_status(dev)
    print(f"  range guard: {st}")


def _x2_backward(dev, meta, table, ws, x7t, pts, radius, eps, d_sdf, H, N2, reroute, n_active=16):
    """One launch of the shipped backward through the C ABI: gather -> x2 image, rsdf_sdfmlp_fd7_bwd_x2 -> d_planes and the
    weight gradients."""
    from rise_sdf_amd import _lib
    L = _lib.lib()
    S = x7t.shape[1]
    p = _lib.ptr
    st = _lib.stream_ptr()
    x2 = torch.empty(int(L.rsdf_x2_bytes(S, 2)), dtype=torch.uint8, device=dev)
    _lib.check(L.rsdf_hashgrid_fwd_fd7_x2(None, p(pts), radius, eps, p(table.detach()), ctypes.byref(meta), S, n_active, 2.0, -1.0, 2,
                                          p(x2), st), "gather")
    flat = [t.detach().contiguous() for wb in ws for t in wb]
    d_planes = torch.full((16, 7, S, 2), float("nan"), device=dev)
    grads = [torch.zeros_like(t) for t in flat]
    guard = torch.empty(8, dtype=torch.int32, device=dev)
    x7s = torch.empty(7, S, 3, device=dev) if reroute else None
    _lib.check(L.rsdf_sdfmlp_fd7_bwd_x2(p(x2), 2, 16, n_active, H, N2, *[p(t) for t in flat], S, p(d_sdf), None, None, p(guard),
                                        p(x7s), reroute, p(d_planes), *[p(t) for t in grads], p(_lib.status(dev)), st), "bwd_x2")
    torch.cuda.synchronize()
    return d_planes, grads, guard.tolist()


def _round3_backward(dev, meta, table, ws, x7t, pts, radius, eps, d_sdf, H, N2, n_active=16):
    from rise_sdf_amd import _lib
    L = _lib.lib()
    S = x7t.shape[1]
    p = _lib.ptr
    st = _lib.stream_ptr()
    planes = torch.zeros(16, 7, S, 2, device=dev)
    _lib.check(L.rsdf_hashgrid_fwd_fd7_pts(p(pts), radius, eps, p(table.detach()), ctypes.byref(meta), S, n_active, p(planes), st), "gather")
    flat = [t.detach().contiguous() for wb in ws for t in wb]
    d_planes = torch.full((16, 7, S, 2), float("nan"), device=dev)
    grads = [torch.zeros_like(t) for t in flat]
    _lib.check(L.rsdf_sdfmlp_fd7_bwd(p(x7t), p(planes), 16, n_active, 2.0, -1.0, H, N2, *[p(t) for t in flat], S, p(d_sdf), None, None,
                                     p(d_planes), *[p(t) for t in grads], st), "bwd round 3")
    torch.cuda.synchronize()
    return d_planes, grads


@pytest.mark.parametrize("H", [64, 128])
def test_outlier_gradient_row_does_not_flush_the_others(dev, ops, H):
    """A single d_sdf row at 2^30 times the typical magnitude sets the launch scale of the x2 gradient images: unguarded,
    every ordinary row's d(hash features) is flushed (shown first, reroute = 0); guarded (the default), the launch is rerouted
    on the device to the range-free kernels and every row agrees with them at 1e-4 of the ordinary rows' largest entry."""
    from rise_sdf_amd import _lib
    N2, S = 13, 6000
    meta, table, ws, x7t, pts, radius, eps = _field_inputs(dev, ops, S, H, N2, seed=41)
    g = torch.Generator().manual_seed(5)
    d_sdf = (torch.randn(7, S, generator=g) * 1e-3).to(dev)
    d_sdf[3, 777] = 1e-3 * 2.0 ** 30
    ref, ref_g = _round3_backward(dev, meta, table, ws, x7t, pts, radius, eps, d_sdf, H, N2)
    others = torch.ones(7, S, dtype=torch.bool, device=dev)
    others[3, 777] = False
    scale = float(ref[:, others].abs().max())
    assert scale > 0
    # unguarded: the ordinary rows are lost
    raw, _, _ = _x2_backward(dev, meta, table, ws, x7t, pts, radius, eps, d_sdf, H, N2, reroute=0)
    lost = float((raw[:, others] - ref[:, others]).abs().max()) / scale
    assert lost > 1e-2, f"the outlier no longer flushes the ordinary rows ({lost:.2e}): is this test still testing anything?"
    # guarded: the launch runs on the range-free kernels
    _lib.poll_status(dev, raise_on_error=False)
    before = _lib.status_totals()
    got, got_g, guard = _x2_backward(dev, meta, table, ws, x7t, pts, radius, eps, d_sdf, H, N2, reroute=1)
    after = _lib.poll_status(dev)
    assert guard[4] == 1 and after["x2_bwd_rerouted"] - before["x2_bwd_rerouted"] == 1, (guard, before, after)
    err = float((got[:, others] - ref[:, others]).abs().max()) / scale
    print(f"H = {H}: ordinary rows vs the range-free kernels: unguarded {lost:.2e}, guarded {err:.2e} (guard words {guard})")
    assert err < 1e-4
    assert float((got[:, 3, 777] - ref[:, 3, 777]).abs().max()) < 1e-4 * float(ref[:, 3, 777].abs().max())
    for a, b in zip(got_g, ref_g):
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max()) + 1e-30


@pytest.mark.parametrize("H", [64, 128])
def test_ordinary_launch_is_not_rerouted_and_forced_reroute_equals_round3(dev, ops, H):
    """(a) gradients spread over six decades (the spread measured inside bench chunks) leave the decision word
    clear; (b) reroute = 2 runs the range-free route unconditionally: d_planes and weight gradients equal the round-3 entry
    point's on fp32 planes to the 2^-24 of the image's hi + lo reconstruction."""
    from rise_sdf_amd import _lib
    N2, S = 13, 5000
    meta, table, ws, x7t, pts, radius, eps = _field_inputs(dev, ops, S, H, N2, seed=43)
    g = torch.Generator().manual_seed(6)
    mag = 10.0 ** (torch.rand(7, S, generator=g) * 6.0 - 6.0)
    d_sdf = (torch.randn(7, S, generator=g).sign() * mag).to(dev)
    ref, ref_g = _round3_backward(dev, meta, table, ws, x7t, pts, radius, eps, d_sdf, H, N2)
    _lib.poll_status(dev, raise_on_error=False)
    before = _lib.status_totals()
    got, got_g, guard = _x2_backward(dev, meta, table, ws, x7t, pts, radius, eps, d_sdf, H, N2, reroute=1)
    after = _lib.poll_status(dev)
    assert guard[4] == 0 and guard[2] == 7 * S and after["x2_bwd_rerouted"] == before["x2_bwd_rerouted"]
    assert after["x2_bwd_guarded"] - before["x2_bwd_guarded"] == 1
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) < 3e-5 * scale
    forced, forced_g, guard2 = _x2_backward(dev, meta, table, ws, x7t, pts, radius, eps, d_sdf, H, N2, reroute=2)
    assert guard2[4] == 1
    assert float((forced - ref).abs().max()) < 2e-6 * scale, float((forced - ref).abs().max()) / scale
    for a, b in zip(forced_g, ref_g):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-30


def test_forced_reroute_through_the_autograd_node(dev, ops, monkeypatch):
    """RSDF_X2_REROUTE=force through rise_sdf_amd.fused.sdf_field_fd7 (the in-place planes, the x7t workspace, the feature
    branch's dh2c): table and weight gradients equal RSDF_X2=0."""
    from rise_sdf_amd import fused
    H, N2, S = 64, 13, 4133
    meta, table, ws, x7t, pts, radius, eps = _field_inputs(dev, ops, S, H, N2, seed=47)
    outs = {}
    for name, env in (("r3", {"RSDF_X2": "0"}), ("forced", {"RSDF_X2_REROUTE": "force"})):
        monkeypatch.delenv("RSDF_X2", raising=False)
        monkeypatch.delenv("RSDF_X2_REROUTE", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for t in [table] + [p for wb in ws for p in wb]:
            t.grad = None
        sdf7t, feat = fused.sdf_field_fd7(x7t, table, ws, meta, 16, 2.0, -1.0, eps / (2 * radius), want_feature=True, points=pts,
                                          radius=radius, eps=eps)
        gs = torch.randn(sdf7t.shape, generator=torch.Generator().manual_seed(1)).to(dev)
        gf = torch.randn(feat.shape, generator=torch.Generator().manual_seed(2)).to(dev)
        ((sdf7t * gs).sum() + (feat * gf).sum()).backward()
        outs[name] = [t.grad.clone() for t in [table] + [p for wb in ws for p in wb]]
    for n, a, b in zip(["table", "w0", "b0", "w1", "b1", "w2", "b2"], outs["r3"], outs["forced"]):
        scale = float(a.abs().max())
        assert scale > 0 and float((a - b).abs().max()) < 1e-5 * scale, (n, float((a - b).abs().max()) / scale)


def test_forward_range_violation_reroutes_to_the_range_free_kernels(dev, ops, monkeypatch):
    """A hidden weight of 5000 (weight_g is a free parameter; the reference's fp32 network stays finite,
    models/network_utils.py:109-157) overflows the x2 format: the forward counts it on the device, the next host read
    switches the SDF field to the range-free kernels with a warning, and a caller that still holds the inputs gets the
    reference's finite values (a 40k-step run must not die where the reference continues).  Inside the
    range nothing trips; RSDF_RANGE_ERROR=raise keeps round 5's named error."""
    import warnings
    import rise_sdf_amd as R
    from rise_sdf_amd import _lib, fused
    monkeypatch.delenv("RSDF_CHECK", raising=False)
    monkeypatch.delenv("RSDF_X2", raising=False)
    monkeypatch.delenv("RSDF_RANGE_ERROR", raising=False)
    for H in (64, 128):
        N2, S = 13, 1000
        meta, table, ws, x7t, pts, radius, eps = _field_inputs(dev, ops, S, H, N2, seed=33)
        call = lambda tb=table: fused.sdf_field_fd7(x7t, tb, ws, meta, 16, 2.0, -1.0, eps / (2 * radius), want_feature=True,   # noqa: E731
                                                    points=pts, radius=radius, eps=eps)
        with torch.no_grad():
            _lib.reset_range_free()
            ws[1][0][3, 5] = 400.0         # large and in range: activations reach ~300 of the 454 the format holds
            call()
            r = R.check_status(dev)
            assert r["x2_fwd_nonfinite"] == 0 and not r["rerouted_now"] and not _lib.range_free("x2")
            ws[1][0][3, 5] = 5000.0
            sdf_bad, _ = call()
            assert not bool(torch.isfinite(sdf_bad).all())
            # ---- the guard closed around the call: finite, and exactly what RSDF_X2=0 computes
            _lib.poll_status(dev, raise_on_error=False)
            assert fused.x2_parts(35, H, N2, "fp32") == 2
            with warnings.catch_warnings(record=True) as wlist:
                warnings.simplefilter("always")
                sdf_ok, feat_ok = R.guarded(call, dev)
            assert len(wlist) == 1 and "range-free" in str(wlist[0].message) and "1023" in str(wlist[0].message)
            assert _lib.range_free("x2") and fused.x2_parts(35, H, N2, "fp32") == 0
            assert bool(torch.isfinite(sdf_ok).all()) and bool(torch.isfinite(feat_ok).all())
            monkeypatch.setenv("RSDF_X2", "0")
            sdf_ref, feat_ref = call()
            monkeypatch.delenv("RSDF_X2")
            assert torch.equal(sdf_ok, sdf_ref) and torch.equal(feat_ok, feat_ref)
            # ... and against the reference's plain fp32 chain (fp64 here) the rerouted values are ordinary fp32 results
            assert _lib.status_totals()["range_reroutes"] >= 1
            # once switched, nothing counts any more and nothing warns again
            with warnings.catch_warnings(record=True) as wlist:
                warnings.simplefilter("always")
                call()
                assert R.check_status(dev)["x2_fwd_nonfinite"] == 0 and not wlist
            # ---- RSDF_RANGE_ERROR=raise: the named error, at an explicit check and behind the marcher's sample-count read
            _lib.reset_range_free()
            monkeypatch.setenv("RSDF_RANGE_ERROR", "raise")
            call()
            with pytest.raises(_lib.RiseSdfHipError, match="RSDF_X2=0"):
                R.check_status(dev)
            assert R.check_status(dev)["x2_fwd_nonfinite"] == 0          # reported once
            call()
            rays = camera_rays(8, 8, seed=1).to(dev)
            from rise_sdf_amd import ops as O
            box = torch.tensor([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5], device=dev)
            tmin, tmax = O.ray_aabb_intersect(rays[:, :3].contiguous(), rays[:, 3:].contiguous(), box)[:2]
            with pytest.raises(_lib.RiseSdfHipError, match="1023"):
                O.march(rays[:, :3].contiguous(), rays[:, 3:].contiguous(), tmin, tmax, box,
                        torch.ones(1, 1, 1, dtype=torch.bool, device=dev), 0.05, 0.0)
            # a table value beyond the input range (|hash feature| >= 255)
            ws[1][0][3, 5] = 1.0
            t2 = table.detach().clone()
            t2[12345] = 300.0
            t2[::7] = 300.0
            call(t2)
            with pytest.raises(_lib.RiseSdfHipError, match="255"):
                R.check_status(dev)
            monkeypatch.delenv("RSDF_RANGE_ERROR")
            # ... rerouted by default: the range-free kernels take fp32 planes, any table magnitude
            sdf_t2, _ = R.guarded(lambda: call(t2), dev)
            assert _lib.range_free("x2") and bool(torch.isfinite(sdf_t2).all())
    _lib.reset_range_free()


def test_sampler_recomputes_an_overflowed_alpha_fn(dev, monkeypatch):
    """The visibility-pruned sampler is the first forward of a step and makes a host read: when its alpha_fn leaves the x2
    range, the poll behind that read switches the field to the range-free kernels and the sampler evaluates alpha_fn again
    -- exact path and capacity mode -- so that the sample set is the one RSDF_X2=0 produces (the reference's fp32 network
    just continues).  A training step after that runs on the range-free kernels."""
    import warnings
    import rise_sdf_amd as R
    from rise_sdf_amd import _lib
    monkeypatch.delenv("RSDF_X2", raising=False)
    monkeypatch.delenv("RSDF_RANGE_ERROR", raising=False)
    sys.path.insert(0, ROOT)
    import bench
    torch.manual_seed(0)
    cfg = bench.c1_config(hidden=64)
    cfg["num_samples_per_ray"] = 128
    cfg["prune_by_visibility"] = True
    model = R.make("neus", cfg).to(dev)
    model.train()
    model.geometry.update_step(0, 0)
    with torch.no_grad():
        model.geometry.network.layers[2].weight_g[7] = 3.0e5          # effective |weight| row norm 3e5: beyond 1023
    rays = camera_rays(24, 24, seed=3).to(dev)
    ro, rd = rays[:, :3].contiguous(), rays[:, 3:].contiguous()
    u = torch.rand(rays.shape[0], generator=torch.Generator().manual_seed(4)).to(dev)
    kw = dict(render_step_size=model.render_step_size, stratified_u=u, cone_angle=0.0, alpha_thre=0.0)
    monkeypatch.setenv("RSDF_X2", "0")
    ref = model.occupancy_grid.sampling(ro, rd, alpha_fn=model._alpha_fn(ro, rd), **kw)
    monkeypatch.delenv("RSDF_X2")
    assert ref[0].numel() > 1000
    for capacity in (False, True):
        _lib.reset_range_free()
        _lib.poll_status(dev, raise_on_error=False)
        grid = model.occupancy_grid
        grid.capacity_mode = capacity
        if capacity:           # a first call sizes the buffers (on the range-free kernels, so that it does not trip the guard)
            monkeypatch.setenv("RSDF_X2", "0")
            grid.sampling(ro, rd, alpha_fn=model._alpha_fn(ro, rd), **kw)
            monkeypatch.delenv("RSDF_X2")
        with warnings.catch_warnings(record=True) as wlist:
            warnings.simplefilter("always")
            got = grid.sampling(ro, rd, alpha_fn=model._alpha_fn(ro, rd), **kw)
        assert any("range-free" in str(w.message) for w in wlist), [str(w.message)[:80] for w in wlist]
        assert _lib.range_free("x2")
        for a, b in zip(got, ref):
            assert torch.equal(a, b), capacity
    grid.capacity_mode = False
    # the model keeps working: forward + backward on the range-free kernels, finite everywhere
    out = model.forward_(rays, stratified_u=u)
    (out["opacity"].sum() + out["depth"].sum()).backward()
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(p.grad).all()) for p in model.parameters() if p.grad is not None)
    assert R.check_status(dev)["x2_fwd_nonfinite"] == 0
    _lib.reset_range_free()


def _saturated_alphas(n_rays, seed):
    """Per-ray alpha profiles of a sharp field: ~1e-5 in front of the surface, a few transition samples, then values at and
    within a few ulps of 1, then the 1e-5 / 1e-5 = 1 plateau inside the object; ragged ray lengths incl. empty rays."""
    g = torch.Generator().manual_seed(seed)
    counts = torch.randint(0, 300, (n_rays,), generator=g)
    counts[::17] = 0
    counts[5] = 1
    counts[6] = 64
    counts[7] = 65
    chunks = []
    for c in counts.tolist():
        k = torch.arange(c, dtype=torch.float32)
        hit = float(torch.randint(5, 200, (1,), generator=g))
        x = (k - hit) * float(torch.rand(1, generator=g) * 6 + 0.5)
        a = torch.sigmoid(x) * (1 - 1e-5) + 1e-5 * torch.rand(c, generator=g)
        a = torch.where(x > 12, torch.ones_like(a) - (torch.randint(0, 3, (c,), generator=g).float() * 2.0 ** -24), a)
        chunks.append(a.clamp(0, 1))
    starts = torch.cumsum(counts, 0) - counts
    packed = torch.stack([starts, counts], dim=1).to(torch.int32)
    return packed, torch.cat(chunks) if chunks else torch.zeros(0)


@pytest.mark.parametrize("seed", [0, 1])
def test_weight_from_alpha_is_the_references_order_of_operations_bit_for_bit(dev, ops, seed, monkeypatch):
    """C1 (lib/nerfacc/cuda/csrc/render_weight.cu:86-153) in the saturated regime: the reference's backward divides the
    rounding residue of its own running subtraction by max(1 - alpha, 1e-10), so the result is a function of its ORDER of
    operations AND of the compiler's multiply-add contraction (the reference binary is nvcc -O3 with the default
    --fmad=true).  The HIP kernels keep that order (one lane per ray for the arithmetic) in both sequences -- contracted (the
    default) and one rounding per source operation (RSDF_C1_FMAD=0) -- and must equal the oracle's restatement of each bit for
    bit: weights, transmittance, the visibility mask and d(alpha), including the 1e3-1e4 entries.  The two sequences
    themselves differ there (shown), which is why the default follows the binary."""
    packed, alphas = _saturated_alphas(700, seed)
    n_rays = packed.shape[0]
    assert int((alphas == 1.0).sum()) > 1000
    gw = torch.randn(alphas.shape, generator=torch.Generator().manual_seed(seed + 10))
    grads = {}
    for fmad in (True, False):
        monkeypatch.setattr(oracle, "C1_FMAD", fmad)
        monkeypatch.setenv("RSDF_C1_FMAD", "1" if fmad else "0")
        a_ref = alphas.clone().requires_grad_(True)
        w_ref, t_ref = oracle.render_weight_from_alpha(a_ref, packed_info=packed)
        w_ref.backward(gw)
        a_hip = alphas.to(dev).requires_grad_(True)
        w, t = ops.render_weight_from_alpha(a_hip, packed_info=packed.to(dev))
        w.backward(gw.to(dev))
        assert torch.equal(w.detach().cpu(), w_ref.detach()) and torch.equal(t.detach().cpu(), t_ref.detach())
        big = float(a_ref.grad.abs().max())
        print(f"  fmad {fmad}: largest |d_alpha| {big:.3g} (the closed form is bounded by |gw| ~ 4)")
        assert big > 1e2, "the residue amplifier is not exercised"
        assert torch.equal(a_hip.grad.cpu(), a_ref.grad), fmad
        grads[fmad] = a_ref.grad.clone()
    d = (grads[True] - grads[False]).abs()
    print(f"  contracted vs uncontracted d_alpha: {int((d > 0).sum())} of {d.numel()} entries differ, by up to {float(d.max()):.3g}")
    assert float(d.max()) > 1.0, "the two sequences agree: the saturated regime is not exercised"
    keep = ops.render_visibility(alphas.to(dev), packed_info=packed.to(dev), early_stop_eps=1e-4, alpha_thre=0.0)
    keep_ref = oracle.render_visibility(alphas, packed_info=packed, early_stop_eps=1e-4)
    assert torch.equal(keep.cpu().bool(), keep_ref)
