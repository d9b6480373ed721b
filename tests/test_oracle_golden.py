"""CPU: pin the oracle against (a) the reference's docstring known-answer vectors and (b) golden
vectors generated from the imported reference Python (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

import oracle


def load(golden_dir, name):
    return {k: (torch.tensor(v) if v.shape != () else v) for k, v in np.load(os.path.join(golden_dir, name)).items()}


# ---- docstring KATs: lib/nerfacc/vol_rendering.py:303-307, 430-434, 493-500 ----------------------------
def test_compositing_known_answers():
    a = torch.tensor([0.4, 0.8, 0.1, 0.8, 0.1, 0.0, 0.9])
    ri = torch.tensor([0, 0, 0, 1, 1, 2, 2])
    t = oracle.render_transmittance_from_alpha(a, ray_indices=ri, n_rays=3)
    assert torch.allclose(t, torch.tensor([1.0, 0.6, 0.12, 1.0, 0.2, 1.0, 1.0]), atol=1e-7)
    w, _ = oracle.render_weight_from_alpha(a, ray_indices=ri, n_rays=3)
    assert torch.allclose(w, torch.tensor([0.4, 0.48, 0.012, 0.8, 0.02, 0.0, 0.9]), atol=1e-7)
    vis = oracle.render_visibility(a, ray_indices=ri, n_rays=3, early_stop_eps=0.3, alpha_thre=0.2)
    assert vis.tolist() == [True, True, False, True, False, False, True]


def test_weight_backward_matches_autograd_of_definition():
    g = torch.Generator().manual_seed(0)
    counts = torch.tensor([5, 0, 9, 1, 30])
    ri = torch.repeat_interleave(torch.arange(5), counts)
    a = (torch.rand(ri.numel(), generator=g, dtype=torch.float64) * 0.6).requires_grad_(True)
    gw = torch.randn(ri.numel(), generator=g, dtype=torch.float64)
    # definition in fp64 torch: T_i = prod_{j<i}(1-a_j)
    ws = []
    start = 0
    for c in counts.tolist():
        seg = a[start:start + c]
        T = torch.cat([torch.ones(1, dtype=torch.float64), torch.cumprod(1 - seg, 0)[:-1]]) if c else seg
        ws.append(seg * T)
        start += c
    (torch.cat(ws) * gw).sum().backward()
    a32 = a.detach().float().requires_grad_(True)
    w, _ = oracle.render_weight_from_alpha(a32, ray_indices=ri, n_rays=5)
    (w * gw.float()).sum().backward()
    assert torch.allclose(w.double(), torch.cat(ws).detach(), atol=1e-6)
    assert torch.allclose(a32.grad.double(), a.grad, rtol=1e-4, atol=1e-5)


# ---- golden: VanillaMLP (models/network_utils.py:109-157) ------------------------------------------------
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_vanilla_mlp_matches_reference(golden_dir, tag):
    z = load(golden_dir, f"vanilla_mlp_{tag}.npz")
    params = [{"g": z[f"layers_{i}_weight_g"].clone().requires_grad_(True),
               "v": z[f"layers_{i}_weight_v"].clone().requires_grad_(True),
               "b": z[f"layers_{i}_bias"].clone().requires_grad_(True)} for i in (0, 2, 4)]
    x = z["x"].clone().requires_grad_(True)
    y = oracle.vanilla_mlp(x, params)
    assert torch.allclose(y, z["y"], rtol=1e-6, atol=1e-6)
    (y * z["gy"]).sum().backward()
    assert torch.allclose(x.grad, z["gx"], rtol=1e-5, atol=1e-6)
    for i, p in zip((0, 2, 4), params):
        for k, n in (("g", "weight_g"), ("v", "weight_v"), ("b", "bias")):
            ref = z[f"grad_layers_{i}_{n}"]
            assert torch.allclose(p[k].grad, ref, rtol=1e-4, atol=1e-5 * float(ref.abs().max() + 1e-9)), (i, n)


def test_relu_mlp_matches_reference(golden_dir):
    z = load(golden_dir, "vanilla_mlp_relu.npz")
    params = [{"w": z[f"layers_{i}_weight"], "b": z[f"layers_{i}_bias"]} for i in (0, 2, 4)]
    assert torch.allclose(oracle.vanilla_mlp(z["x"], params, activation="relu"), z["y"], rtol=1e-6, atol=1e-6)


def test_sphere_init_statistics():
    """The oracle's own initialiser follows network_utils.py:130-144 (structure, not RNG stream)."""
    p = oracle.sphere_init_mlp_params(35, 48, 64, 2, radius=0.5, seed=0)
    assert float(p[0]["v"][:, 3:].abs().max()) == 0.0
    assert torch.allclose(p[2]["b"], torch.full((48,), -0.5))
    assert abs(float(p[2]["v"].mean()) - (np.pi ** 0.5) / 8.0) < 1e-3
    for q in p:  # weight_norm wrap: g == ||v|| so that W == v at init
        assert torch.allclose(oracle.weight_norm_effective(q["g"], q["v"]), q["v"], atol=1e-6)


# ---- golden: get_alpha (models/split_mixed_occ.py:151-177) ------------------------------------------------
def test_get_alpha_matches_reference(golden_dir):
    z = load(golden_dir, "get_alpha.npz")
    for vi, v in enumerate(z["variances"].tolist()):
        for ci, c in enumerate(z["cos_anneal"].tolist()):
            a = oracle.get_alpha(z["sdf"], z["normal"], z["dirs"], z["dists"],
                                 oracle.inv_s_from_variance(torch.tensor(v, dtype=torch.float32)), c)
            assert torch.equal(a, z[f"alpha_v{vi}_c{ci}"]), (vi, ci)


# ---- golden: VolumeSDF finite-difference forward/backward (models/geometry.py:206-244) ---------------------
def test_volume_sdf_fd_matches_reference(golden_dir):
    z = load(golden_dir, "volume_sdf_fd.npz")
    meta, n_params = oracle.grid_meta(6, 2, 12, 8, 1.5)
    assert n_params == z["p__encoding_encoding_encoding_params"].numel()
    for step in z["steps"].tolist():
        table = z["p__encoding_encoding_encoding_params"].clone().requires_grad_(True)
        mlp = [{"g": z[f"p__network_layers_{i}_weight_g"].clone().requires_grad_(True),
                "v": z[f"p__network_layers_{i}_weight_v"].clone().requires_grad_(True),
                "b": z[f"p__network_layers_{i}_bias"].clone().requires_grad_(True)} for i in (0, 2, 4)]
        level = oracle.progressive_level(step, 3, 0, 100, 6)
        assert level == int(z[f"s{step}_level"])
        eps = oracle.progressive_fd_eps(1.5, 8, 1.5, level)
        assert abs(eps - float(z[f"s{step}_eps"])) < 1e-12
        sdf, grad, feat = oracle.volume_sdf(z["pts"], table, meta, mlp, radius=1.5, fd_eps=eps,
                                            n_active_levels=level)
        assert torch.allclose(sdf, z[f"s{step}_sdf"], rtol=1e-6, atol=1e-6)
        assert torch.allclose(feat, z[f"s{step}_feature"], rtol=1e-6, atol=1e-6)
        assert torch.allclose(grad, z[f"s{step}_grad"], rtol=1e-5, atol=1e-5)
        loss = (sdf * z[f"s{step}_gs"]).sum() + (grad * z[f"s{step}_gg"]).sum() + (feat ** 2).sum() * 0.1
        loss.backward()
        ref = z[f"s{step}_grad__encoding_encoding_encoding_params"]
        assert torch.allclose(table.grad, ref, rtol=1e-3, atol=1e-4 * float(ref.abs().max()))
        for i, p in zip((0, 2, 4), mlp):
            for k, n in (("g", "weight_g"), ("v", "weight_v"), ("b", "bias")):
                ref = z[f"s{step}_grad__network_layers_{i}_{n}"]
                assert torch.allclose(p[k].grad, ref, rtol=1e-3, atol=1e-4 * float(ref.abs().max() + 1e-9))


# ---- golden: rays (models/ray_utils.py:9-56) ---------------------------------------------------------------
def test_rays_match_reference(golden_dir):
    z = load(golden_dir, "rays.npz")
    W, H, focal = int(z["W"]), int(z["H"]), float(z["focal"])
    dirs = oracle.get_ray_directions(W, H, focal, focal, W / 2, H / 2)
    assert torch.equal(dirs, z["directions"])
    ro, rd = oracle.get_rays(dirs, z["c2w"])
    assert torch.equal(ro, z["rays_o"]) and torch.allclose(rd, z["rays_d"], rtol=0, atol=0)


# ---- golden: rendering orchestration (models/volrend.py:739-895) ---------------------------------------------
def test_rendering_orchestration_matches_reference(golden_dir):
    z = load(golden_dir, "rendering.npz")
    ri, a = z["ray_indices"], z["alphas"]
    w, t = oracle.render_weight_from_alpha(a, ray_indices=ri, n_rays=5)
    assert torch.allclose(w, z["weights"], atol=1e-7) and torch.allclose(t, z["trans"], atol=1e-7)
    acc = lambda v: oracle.accumulate_along_rays(w, v, ray_indices=ri, n_rays=5)
    assert torch.allclose(acc(z["rgbs"]), z["colors"], atol=1e-6)
    assert torch.allclose(acc(z["normals_in"]), z["normals"], atol=1e-6)
    assert torch.allclose(acc(None), z["opacities"], atol=1e-6)
    mid = ((z["t_starts"] + z["t_ends"]) / 2.0)[:, None]
    assert torch.allclose(acc(mid), z["depths"], atol=1e-6)  # depth is NOT normalised by opacity


# ---- hash grid: level table of SURVEY.md Appendix B and internal consistency --------------------------------
def test_grid_meta_matches_survey_table():
    m, n = oracle.grid_meta(16, 2, 19, 32, 1.447269237440378)
    assert n == 14533536
    assert list(m.res)[:16] == [32, 47, 68, 98, 141, 204, 295, 426, 616, 892, 1291, 1868, 2703, 3912, 5661, 8192]
    assert list(m.offset)[:4] == [0, 32768, 136592, 451024]
    m16, n16 = oracle.grid_meta(16, 2, 19, 16, 1.447269237440378)
    assert n16 == 12599920
    m4, n4 = oracle.grid_meta(4, 2, 19, 16, 1.447269237440378)
    assert n4 == 174880 * 2


def test_hashgrid_is_trilinear_and_gradient_is_adjoint():
    meta, n = oracle.grid_meta(4, 2, 10, 4, 2.0)
    g = torch.Generator().manual_seed(3)
    table = torch.randn(n, generator=g)
    # interpolation property: at a grid vertex of level 0 (dense) the feature equals the table entry
    scale0, res0 = meta.scale[0], meta.res[0]
    ijk = torch.tensor([[1, 2, 0], [0, 0, 0], [2, 1, 2]])
    x = (ijk.float() - 0.5 + 1e-6) / scale0 + 0.0  # pos = x*scale + 0.5 ~= ijk
    x = ((ijk.float() - 0.5) / scale0).clamp(0, 1)
    out = oracle.hashgrid_encode(x, table, meta)
    for r, (i, j, k) in enumerate(ijk.tolist()):
        if min(i, j, k) >= 1:  # x >= 0 needs ijk >= 0.5 -> first row only
            e = i + j * res0 + k * res0 * res0
            assert torch.allclose(out[r, :2], table[2 * e:2 * e + 2], atol=1e-5)
    # adjoint test: <enc(x; T), G> == <T, grad>
    x = torch.rand(200, 3, generator=g)
    G = torch.randn(200, 8, generator=g)
    T = table.clone().requires_grad_(True)
    (oracle.hashgrid_encode(x, T, meta) * G).sum().backward()
    lhs = float((oracle.hashgrid_encode(x, table, meta) * G).sum())
    rhs = float((table * T.grad).sum())
    assert abs(lhs - rhs) < 1e-3 * (abs(lhs) + 1)
    idx = oracle.hashgrid_indices(x, meta)
    assert int(idx.min()) >= 0 and int(idx.max()) < n // 2


def test_weight_backward_contraction_modes(monkeypatch):
    """render_weight.cu:139-151 as the nvcc default (--fmad=true) contracts it vs one rounding per source operation
    (oracle.C1_FMAD): identical where the chain is well conditioned (both within fp32 rounding of the closed form),
    different in the saturated regime, where d_alpha is the amplified rounding residue of the running subtraction."""
    g = torch.Generator().manual_seed(0)
    n_rays, steps = 64, 48
    packed = torch.stack([torch.arange(n_rays) * steps, torch.full((n_rays,), steps)], 1).to(torch.int32)
    gw = torch.randn(n_rays * steps, generator=g)

    def grad(alphas, fmad):
        monkeypatch.setattr(oracle, "C1_FMAD", fmad)
        a = alphas.clone().requires_grad_(True)
        w, _ = oracle.render_weight_from_alpha(a, packed_info=packed)
        w.backward(gw)
        return a.grad

    soft = torch.rand(n_rays * steps, generator=g) * 0.2
    a, b = grad(soft, True), grad(soft, False)
    assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())
    hard = soft.clone().view(n_rays, steps)
    hard[:, 20] = 1.0                                     # a saturated sample: everything behind it has T = 0
    hard = hard.reshape(-1)
    a, b = grad(hard, True), grad(hard, False)
    assert float(a.abs().max()) > 1e2 and float(b.abs().max()) > 1e2        # the residue / 1e-10
    assert float((a - b).abs().max()) > 1.0
    # in front of the saturated sample both agree with the closed form d w_k / d alpha_j to rounding
    front = torch.arange(n_rays * steps).view(n_rays, steps)[:, :20].reshape(-1)
    assert float((a[front] - b[front]).abs().max()) <= 1e-4 * float(a[front].abs().max())
