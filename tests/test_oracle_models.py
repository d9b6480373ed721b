"""The oracle's restatement of the ASSEMBLED split-mixed-occ model (oracle/split_mixed_occ.py) against outputs of the
reference's own ``SplitMixedOCCModel.forward_`` (tests/golden/models_split_mixed_occ.npz, written by
tests/golden/make_golden_models.py from the imported reference with its CUDA-only callees filled by the oracle): pins the
orchestration of models/split_mixed_occ.py:179-222,224-443 -- sampling with visibility pruning, channel slicing, the
secondary-ray blend, the relighting third bounce, background compositing and sRGB -- which round 2 only restated."""
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import split_mixed_occ as OS
from oracle import texture as OT
from helpers import sphere_binary

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "models_split_mixed_occ.npz")
CASES = {"s0": (0, False, False), "s0_indirect": (0, True, False), "s1_indirect": (1, True, False),
         "s1_relight": (1, True, True)}
KEYS0 = ("comp_rgb", "comp_diffuse_rgb", "comp_spec_rgb", "comp_blend", "comp_normal", "opacity", "depth", "comp_rgb_full")
KEYS1 = KEYS0 + ("comp_rgb_phys", "comp_diffuse_rgb_phys", "comp_spec_rgb_phys", "comp_albedo", "comp_metallic",
                 "comp_roughness", "comp_rgb_phys_full", "comp_spec_rgb_full", "comp_spec_rgb_phys_full")


def load_fixture():
    z = np.load(GOLDEN)
    return {k: torch.from_numpy(z[k]) for k in z.files}


def relight_base():
    return torch.rand(6, 64, 64, 3, generator=torch.Generator().manual_seed(9)) * 2.0


BIG_GOLDEN = os.path.join(os.path.dirname(GOLDEN), "models_split_mixed_occ_l16_h128.npz")


def load_big_fixture():
    """models_split_mixed_occ_l16_h128.npz plus what it regenerates from seeds (tests/golden/make_golden_models.py
    big_table / big_envmap): the 58 MB hash table and the environment map."""
    z = np.load(BIG_GOLDEN)
    fx = {k: torch.from_numpy(z[k]) for k in z.files}
    g = torch.Generator().manual_seed(int(fx["table_seed"]))
    fx["p__geometry.encoding.encoding.encoding.params"] = (torch.rand(int(fx["n_table"]), generator=g) * 2 - 1) * 1e-3
    fx["p__emitter.base"] = torch.rand(6, 64, 64, 3, generator=torch.Generator().manual_seed(int(fx["envmap_seed"]))) * 0.5 + 0.25
    return fx


def oracle_params_from_fixture(fx, relighting=False, grad=False, big=False):
    """The fixture's state_dict (the reference's parameter names) as the oracle's plain structures."""
    t = lambda k: fx["p__" + k].clone().requires_grad_(grad)   # noqa: E731
    meta, n_params = oracle.grid_meta(16, 2, 19, 32, 1.447269237440378) if big else oracle.grid_meta(4, 2, 14, 16, 1.5)
    table = t("geometry.encoding.encoding.encoding.params")
    assert table.numel() == n_params
    mlp = [{"g": t(f"geometry.network.layers.{i}.weight_g"), "v": t(f"geometry.network.layers.{i}.weight_v"),
            "b": t(f"geometry.network.layers.{i}.bias")} for i in (0, 2, 4)]
    nets = {}
    for name, idx in (("albedo", (0, 2, 4, 6, 8)), ("roughness", (0, 2, 4)), ("metallic", (0, 2, 4)),
                      ("env", (0, 2, 4, 6, 8)), ("secondary", (0, 2, 4, 6, 8))):
        nets[name] = [{"w": t(f"texture.{name}_network.layers.{i}.weight"), "b": t(f"texture.{name}_network.layers.{i}.bias")}
                      for i in idx]
    shell = [float(v) for v in fx["shell"]]
    return dict(table=table, meta=meta, mlp=mlp, var=t("variance.variance"), nets=nets,
                binary=sphere_binary(128, *shell), radius=1.5, fd_eps=float(fx["fd_eps"]),
                render_step_size=float(fx["render_step_size"]), sec_near=0.05, sec_far=1.5, sec_steps=24,
                background=torch.ones(3), fg_lut=OT.synthetic_fg_lut(int(fx["lut_res"])),
                emitter_base=(relight_base() if relighting else t("emitter.base")), relighting_threshold=0.6)


@pytest.mark.parametrize("tag", list(CASES))
def test_oracle_model_matches_reference_forward(tag):
    fx = load_fixture()
    stage, indirect, relighting = CASES[tag]
    P = oracle_params_from_fixture(fx, relighting)
    with torch.no_grad():
        out = OS.render(fx["rays"], P, stage=stage, indirect=indirect, relighting=relighting, stratified_u=None)
    # the sample sets the reference's run produced (recorded inside its OccGridEstimator stub) are the oracle's own
    for name in ("primary",) + (("secondary",) if indirect else ()):
        for i, part in enumerate(("ri", "ts", "te")):
            assert torch.equal(out["own_" + name][i], fx[f"{tag}__{name}_{part}"]), (name, part)
    for k in (KEYS1 if stage else KEYS0):
        ref = fx[f"{tag}__{k}"]
        # radiance-like outputs at the north star's 1e-4; the relit third bounce re-queries FD normals at points that
        # themselves come from a composited depth (largest gap seen: 6e-5 on one pixel)
        atol = 1e-4 if relighting else 2e-5
        assert torch.allclose(out[k], ref, rtol=1e-4, atol=atol), (tag, k, float((out[k] - ref).abs().max()))
    assert int((fx[f"{tag}__opacity"][:, 0] > 0.5).sum()) == out["valid_indices"].numel() > 30
    if relighting:
        assert int(out["rmask"].sum()) > 0, "the relighting fixture must exercise the third bounce"


TIGHT_L16 = ("comp_diffuse_rgb", "comp_blend", "opacity", "depth", "comp_albedo", "comp_metallic", "comp_roughness",
             "comp_diffuse_rgb_phys")


def check_against_big_fixture(out, fx, who):
    """Gates shared by the oracle (CPU) and the HIP model (GPU) against the reference-run fixture at L = 16 / H = 128.

    At L = 16 the progressive FD eps is one cell of the finest grid (3.7e-4): the normal divides an fp32 SDF difference by
    2 eps, so ANY two fp32 evaluations of the same network -- here torch's CPU GEMMs on other batch shapes -- disagree by up to
    ~1e-3 in a normal component.  In the full PBR model that reaches further than in NeuS: the secondary (reflection) rays start
    at the composited depth along the composited normal, their visibility-pruned sample sets differ in a few borderline
    samples (14 of 312 between the oracle and the reference run), and ``tr`` of such a ray moves the specular term of its pixel
    by up to 0.06 (the oracle) .. 0.4 (the HIP model: other pixels flip).  So: outputs that do not pass through a normal at the
    north star's 1e-4 on all but a few borderline pixels -- a primary sample on the other side of T >= 1e-4 carries a weight of
    a few 1e-4, i.e. up to ~1.5e-3 of depth -- ; the normal map at what fp32 resolves; the specular-dependent outputs within
    1e-3 on all but a bounded number of pixels, median difference below 1e-5."""
    key = lambda r, t: set(zip(r.tolist(), t.contiguous().view(torch.int32).tolist()))   # noqa: E731
    dprim = len(key(*out["own_primary"][:2]) ^ key(fx["primary_ri"], fx["primary_ts"]))
    dsec = len(key(*out["own_secondary"][:2]) ^ key(fx["secondary_ri"], fx["secondary_ts"]))
    print(f"{who}: primary samples differing {dprim} of {fx['primary_ri'].numel()}, secondary {dsec} of {fx['secondary_ri'].numel()}")
    assert dprim <= 6 and dsec <= 40
    rows, failures = [], []
    for k in KEYS1:
        ref, got = fx["out__" + k], out[k]
        d = (got - ref).abs()
        n4 = int((~torch.isclose(got, ref, rtol=1e-4, atol=2e-5).all(-1)).sum())
        n3 = int((d.max(-1).values > 1e-3).sum())
        rows.append(f"{k} max {float(d.max()):.1e} px>1e-4 {n4} px>1e-3 {n3}")
        if k in TIGHT_L16:
            ok = n4 <= 4 + 2 * dprim and float(d.max()) < 2e-3
        elif k == "comp_normal":
            ok = float(d.max()) < 3e-3
        else:       # (a reflected ray that flips between hitting and missing the surface moves its pixel's specular term by
            #         up to the whole difference between the environment and the secondary colour)
            ok = n3 <= 48 and float(d.max()) < 0.5 and float(d.median()) < 1e-5
        if not ok:
            failures.append((k, n4, n3, float(d.max()), float(d.median())))
    print(f"{who} vs the reference's forward_: " + "; ".join(rows))
    assert not failures, (who, failures)


def test_oracle_model_matches_reference_forward_l16_h128():
    """The same at the sizes the shipped kernel family runs (tests/golden/models_split_mixed_occ_l16_h128.npz: L = 16,
    T = 2^19, 2 x 128 SDF network with 48 features, 128-wide radiance networks, stage 1 with secondary-ray occlusion)."""
    fx = load_big_fixture()
    P = oracle_params_from_fixture(fx, big=True)
    with torch.no_grad():
        out = OS.render(fx["rays"], P, stage=1, indirect=True, relighting=False, stratified_u=None)
    assert torch.equal(out["own_primary"][0], fx["primary_ri"]) and torch.equal(out["own_primary"][1], fx["primary_ts"])
    check_against_big_fixture(out, fx, "oracle")
    assert abs(int((fx["out__opacity"][:, 0] > 0.5).sum()) - out["valid_indices"].numel()) <= 1 and out["valid_indices"].numel() > 30


# ---- NeuSModel (models/neus.py:227-317) --------------------------------------------------------------------------------
NEUS_GOLDEN = os.path.join(os.path.dirname(GOLDEN), "models_neus.npz")


def load_neus_fixture():
    z = np.load(NEUS_GOLDEN)
    return {k: torch.from_numpy(z[k]) for k in z.files}


NEUS_BIG_GOLDEN = os.path.join(os.path.dirname(NEUS_GOLDEN), "models_neus_l16_h128.npz")


@pytest.mark.parametrize("big", [False, True])
def test_oracle_neus_matches_reference_forward(big):
    """oracle.neus_geometry_render + the radiance network against the reference's own NeuSModel.forward_ (eval mode,
    occupancy sampling without visibility pruning, volume-radiance texture, white background).  ``big``: the fixture at the
    sizes config[2..4] run (L = 16, T = 2^19, 2 x 128 SDF network, 48 features; table regenerated from its seed)."""
    import torch.nn.functional as F
    if big:
        z = np.load(NEUS_BIG_GOLDEN)
        fx = {k: torch.from_numpy(z[k]) for k in z.files}
        g = torch.Generator().manual_seed(int(fx["table_seed"]))
        fx["p__geometry.encoding.encoding.encoding.params"] = (torch.rand(int(fx["n_table"]), generator=g) * 2 - 1) * 1e-3
    else:
        fx = load_neus_fixture()
    t = lambda k: fx["p__" + k]   # noqa: E731
    meta, n_params = oracle.grid_meta(16, 2, 19, 32, 1.447269237440378) if big else oracle.grid_meta(4, 2, 14, 16, 1.5)
    table = t("geometry.encoding.encoding.encoding.params")
    assert table.numel() == n_params
    mlp = [{"g": t(f"geometry.network.layers.{i}.weight_g"), "v": t(f"geometry.network.layers.{i}.weight_v"),
            "b": t(f"geometry.network.layers.{i}.bias")} for i in (0, 2, 4)]
    tex = [{"w": t(f"texture.network.layers.{i}.weight"), "b": t(f"texture.network.layers.{i}.bias")} for i in (0, 2, 4)]
    rays = fx["rays"]
    roi = OS.ROI(1.5)
    shell = [float(v) for v in fx["shell"]]
    with torch.no_grad():
        ri, ts, te = oracle.ray_marching(rays[:, :3].contiguous(), rays[:, 3:].contiguous(), scene_aabb=roi, grid_roi=roi,
                                         grid_binary=sphere_binary(128, *shell), near_plane=0.0, far_plane=1e10,
                                         render_step_size=float(fx["render_step_size"]))
        assert torch.equal(ri, fx["primary_ri"]) and torch.equal(ts, fx["primary_ts"]) and torch.equal(te, fx["primary_te"])
        ref = oracle.neus_geometry_render(rays, ri, ts, te, table, meta, mlp, t("variance.variance"), radius=1.5,
                                          fd_eps=float(fx["fd_eps"]))
        t_dirs = rays[:, 3:][ri]
        normal = F.normalize(ref["sdf_grad"], p=2, dim=-1)                  # models/neus.py:258
        rgb = torch.sigmoid(OT.relu_mlp(torch.cat([ref["feature"], OT.sh_encode((t_dirs + 1) / 2, 4), normal], -1), tex))
        comp = oracle.accumulate_along_rays(ref["weights"], rgb, ray_indices=ri, n_rays=rays.shape[0])
    got = {"comp_rgb": comp, "opacity": ref["opacity"], "depth": ref["depth"],
           "comp_normal": F.normalize(ref["comp_normal"], p=2, dim=-1),
           "comp_rgb_full": comp + 1.0 * (1.0 - ref["opacity"])}
    for k, v in got.items():
        d = float((v - fx["out__" + k]).abs().max())
        if big and k in ("comp_rgb", "comp_normal", "comp_rgb_full"):
            # At L = 16 the finite-difference eps is one cell of the 2048-grid (3.7e-4): the normal divides an fp32 SDF
            # difference by 2 eps, so even two fp32 evaluations on the SAME CPU that batch the seven taps differently (the
            # reference's forward_ and this oracle) disagree by ~1e-3 in a normal component and 2.5e-4 in the colour
            assert d < (1e-2 if k == "comp_normal" else 1e-3), (k, d)
        else:
            assert torch.allclose(v, fx["out__" + k], rtol=1e-5, atol=2e-6), (k, d)
    assert int(fx["out__num_samples_full"]) == ri.numel() and int((fx["out__opacity"][:, 0] > 0.5).sum()) > (20 if big else 50)
