#!/usr/bin/env python3
"""Golden vectors for the ASSEMBLED models, by running the reference's own classes in the build container
(VERDICT r02 item 7):

    python tests/golden/make_golden_models.py          ->  tests/golden/models_split_mixed_occ.npz, models_neus.npz,
                                                            models_neus_l16_h128.npz, models_split_mixed_occ_l16_h128.npz

The reference's ``SplitMixedOCCModel.forward_`` (models/split_mixed_occ.py:224-443, incl. compute_indirect_radiance
:179-222 and the relighting branch :320-331) and ``NeuSModel.forward_`` (models/neus.py:227-317) are EXECUTED, on the CPU, in
eval mode, with their CUDA-only callees filled by the oracle exactly as tests/golden/make_golden.py does for the texture
and light fixtures:

    tcnn.Encoding                          -> oracle hash grid / SH basis              (tiny-cuda-nn absent: unpinned)
    nerfacc.OccGridEstimator.sampling      -> oracle.ray_marching + visibility pruning (vendored 0.3.5 semantics)
    render_weight_from_alpha, accumulate   -> oracle (pinned by the reference's docstring KATs)
    dr.texture, ru.diffuse/specular_cubemap-> oracle.envlight / oracle.gridsample, fp64 (nvdiffrast absent: unpinned)
    FG LUT file                            -> oracle.texture.synthetic_fg_lut()        (file not in the repository)

What the fixtures pin is therefore the ORCHESTRATION: which tensors go where, channel slicing, the secondary-ray blend,
the third bounce, background compositing and sRGB, the output dictionary -- the reference's code, not a restatement.
tests/test_oracle_models.py checks oracle/split_mixed_occ.py against them on the CPU; tests/test_gpu_model_fixtures.py
checks the HIP models.  Only data is written: rays, parameters (state_dict), outputs.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE)))
import make_golden as mg  # noqa: E402
import oracle  # noqa: E402
from oracle import envlight as oenv, gridsample as ogs, texture as otex  # noqa: E402
from helpers import camera_rays, sphere_binary  # noqa: E402

SHELL = (0.2, 0.9)          # occupancy: sphere_binary(128, *SHELL)
LUT_RES = 64


class OracleOccGrid(torch.nn.Module):
    """nerfacc.OccGridEstimator as the reference uses it (models/split_mixed_occ.py:83-86,200-208,264-272), backed by
    the oracle marcher; records every sampling call's result."""

    def __init__(self, roi_aabb=None, resolution=128, levels=1):
        super().__init__()
        self.roi = torch.as_tensor(roi_aabb, dtype=torch.float32)
        self.binaries = sphere_binary(resolution, *SHELL)[None]
        self.calls = []

    def sampling(self, rays_o, rays_d, sigma_fn=None, alpha_fn=None, near_plane=0.0, far_plane=1e10, t_min=None,
                 t_max=None, render_step_size=1e-3, early_stop_eps=1e-4, alpha_thre=0.0, stratified=False,
                 cone_angle=0.0):
        assert not stratified and sigma_fn is None
        out = oracle.ray_marching(rays_o.contiguous(), rays_d.contiguous(), scene_aabb=self.roi, grid_roi=self.roi,
                                  grid_binary=self.binaries[0], near_plane=near_plane, far_plane=far_plane,
                                  render_step_size=render_step_size, alpha_fn=alpha_fn, early_stop_eps=early_stop_eps,
                                  alpha_thre=alpha_thre)
        self.calls.append(tuple(t.clone() for t in out))
        return out

    def update_every_n_steps(self, *a, **k):
        raise RuntimeError("eval-mode fixtures never update the grid")


def dr_texture(tex, uv, mip=None, mip_level_bias=None, filter_mode="linear", boundary_mode="wrap"):
    if boundary_mode == "cube":
        d = uv.reshape(-1, 3).double()
        if filter_mode == "linear":
            out = oenv.cube_sample_linear(tex[0].double(), d)
        else:
            out = oenv.cube_sample_mip([tex[0].double()] + [m[0].double() for m in mip], d,
                                       mip_level_bias.reshape(-1).double())
        return out.float().reshape(*uv.shape[:-1], -1)
    assert boundary_mode == "clamp" and filter_mode == "linear"
    out = ogs.grid_sample_2d(tex.double().permute(0, 3, 1, 2), uv.double() * 2.0 - 1.0, "border", False)
    return out.permute(0, 2, 3, 1).float()


def nodev(fn):
    def f(*a, **k):
        k.pop("device", None)
        return fn(*a, **k)
    return f


class NoDev:
    def __init__(self, *a):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def model_cfg(indirect, stage1, tex_hidden=64):
    mlp = lambda n: {"otype": "VanillaMLP", "activation": "ReLU", "output_activation": "none", "n_neurons": tex_hidden,   # noqa: E731
                     "n_hidden_layers": n}
    return mg.Cfg({
        "name": "split-mixed-occ", "radius": 1.5, "num_samples_per_ray": 1024, "num_samples_per_secondary_ray": 24,
        "grid_prune": True, "randomized": True, "ray_chunk": 4096, "cos_anneal_end": 0, "learned_background": False,
        "indirect_pred": indirect, "relighting_threshold": 0.6, "split_sum_kick_in_step": 0 if stage1 else 1 << 60,
        "variance": {"init_val": 0.6, "modulate": False},
        "geometry": {"name": "volume-sdf", "radius": 1.5, "feature_dim": 13, "grad_type": "finite_difference",
                     "finite_difference_eps": "progressive", "isosurface": None,
                     "xyz_encoding_config": {"otype": "ProgressiveBandHashGrid", "n_levels": 4, "n_features_per_level": 2,
                                             "log2_hashmap_size": 14, "base_resolution": 16, "per_level_scale": 1.5,
                                             "include_xyz": True, "start_level": 4, "start_step": 0, "update_steps": 1},
                     "mlp_network_config": {"otype": "VanillaMLP", "activation": "ReLU", "output_activation": "none",
                                            "n_neurons": 32, "n_hidden_layers": 2, "sphere_init": True,
                                            "sphere_init_radius": 0.5, "weight_norm": True}},
        "texture": {"name": "volume-mixed-mip-split-occ", "input_feature_dim": 13, "other_dim": 3, "sample_size": 8,
                    "dir_encoding_config": {"otype": "SphericalHarmonics", "degree": 5, "reflected": True},
                    "metallic_mlp_network_config": mlp(2), "albedo_mlp_network_config": mlp(4),
                    "spec_mlp_network_config": mlp(4), "roughness_mlp_network_config": mlp(2),
                    "secondary_mlp_network_config": mlp(4),
                    "xyz_encoding_config": {"otype": "VanillaFrequency", "n_frequencies": 6},
                    "color_activation": "sigmoid"},
        "light": {"name": "envlight-mip-cube",
                  "envlight_config": {"scale": 0.5, "bias": 0.25, "base_res": 64, "hdr_filepath": None}},
    })


def main():
    mg.install_stubs()
    sys.modules["nerfacc"].OccGridEstimator = OracleOccGrid
    for modname in ("nerfacc", "nerfacc.volrend"):
        sys.modules[modname].render_weight_from_alpha = oracle.render_weight_from_alpha
        sys.modules[modname].accumulate_along_rays = oracle.accumulate_along_rays
    import nvdiffrast.torch as dr_stub
    dr_stub.texture = dr_texture
    # device / file patches for the whole run (the reference allocates on get_rank() == cuda:0 and reads the LUT file)
    torch.cuda.device = NoDev
    orig = (torch.zeros, torch.rand, torch.linspace, torch.ones, torch.as_tensor)
    torch.zeros, torch.rand, torch.linspace, torch.ones = (nodev(f) for f in orig[:4])
    np.fromfile = lambda *a, **k: otex.synthetic_fg_lut(LUT_RES).numpy().reshape(-1) \
        if LUT_RES == 256 else np.zeros(256 * 256 * 2, dtype=np.float32)
    import models  # noqa: F401
    import utils.misc as misc
    misc.get_rank = lambda: 0
    from models import network_utils as nu
    nu.config_to_primitive = lambda c: dict(c)
    nu.get_rank = lambda: 0
    from lib.pbr import light as rlight
    from lib.pbr.utils import light_utils as rlu
    rlight.ru.diffuse_cubemap = lambda c: oenv.diffuse_cubemap(c.double()).float()
    rlight.ru.specular_cubemap = lambda c, r, cutoff=0.99: oenv.specular_cubemap(c.double(), r, cutoff).float()
    rlight.dr.texture = dr_texture
    rlu.dr.texture = dr_texture
    from models import texture as rtex
    rtex.dr.texture = dr_texture
    from models.split_mixed_occ import SplitMixedOCCModel

    rays = camera_rays(18, 18, seed=2)
    saved = {"rays": rays, "shell": np.array(SHELL), "lut_res": np.array(LUT_RES)}
    state = None
    if "--only-pbr-l16" in sys.argv:          # (development shortcut: the other fixtures are left untouched)
        return pbr_l16_h128(SplitMixedOCCModel)
    for tag, (indirect, stage1, relighting) in {"s0": (False, False, False), "s0_indirect": (True, False, False),
                                                "s1_indirect": (True, True, False), "s1_relight": (True, True, True)}.items():
        torch.manual_seed(5)
        model = SplitMixedOCCModel(model_cfg(indirect, stage1))
        with torch.no_grad():
            if state is None:
                # a lumpy blob with a sharp surface, as tests/test_gpu_split_model.py: opaque pixels, occluded reflections
                model.geometry.encoding.encoding.encoding.params.mul_(10.0)
                l0 = model.geometry.network.layers[0]
                l0.weight_v[:, 3:] = torch.randn_like(l0.weight_v[:, 3:]) * 0.3
                model.texture.FG_LUT = otex.synthetic_fg_lut(LUT_RES)
                state = {k: v.clone() for k, v in model.state_dict().items() if not k.endswith('FG_LUT')}
            else:
                model.load_state_dict(state, strict=False)
                model.texture.FG_LUT = otex.synthetic_fg_lut(LUT_RES)
        model.eval()
        model.update_step(0, 0)
        assert model.stage == int(stage1)
        model.background_color = torch.ones(3)
        if relighting:      # a different environment, as systems/split_occ.py:405-420 swaps it in
            with torch.no_grad():
                model.emitter.base.copy_(torch.rand(model.emitter.base.shape, generator=torch.Generator().manual_seed(9)) * 2.0)
            # (regenerated from its seed by the tests: not stored)
        with torch.no_grad():
            model.emitter.build_mips()
            out = model.forward_(rays, relighting=relighting)
        calls = model.occupancy_grid.calls
        n_valid = int((out["opacity"][:, 0] > 0.5).sum())
        print(tag, "samples", calls[0][0].numel(), "valid rays", n_valid, "secondary samples",
              calls[1][0].numel() if len(calls) > 1 else 0)
        assert n_valid > 30
        for k, v in out.items():
            if isinstance(v, torch.Tensor) and v.dtype in (torch.float32, torch.bool, torch.int32) and not k.endswith("_bg"):
                saved[f"{tag}__{k}"] = v
        saved[f"{tag}__primary_ri"], saved[f"{tag}__primary_ts"], saved[f"{tag}__primary_te"] = calls[0]
        if len(calls) > 1:
            saved[f"{tag}__secondary_ri"], saved[f"{tag}__secondary_ts"], saved[f"{tag}__secondary_te"] = calls[1]
    for k, v in state.items():
        saved["p__" + k] = v
    saved["fd_eps"] = np.array(model.geometry._finite_difference_eps)
    saved["render_step_size"] = np.array(model.render_step_size)
    mg.save("models_split_mixed_occ.npz", **saved)

    # ---- NeuSModel (models/neus.py:52-92,128-150,227-317): geometry + volume-radiance texture, occupancy sampling
    # WITHOUT visibility pruning, composited colour and the _bg / _full dictionaries ---------------------------------
    from models.neus import NeuSModel
    ncfg = model_cfg(False, False)
    ncfg.update({"name": "neus", "variance": {"init_val": 0.45, "modulate": False}, "num_samples_per_ray": 512,
                 "texture": {"name": "volume-radiance", "input_feature_dim": 13 + 3,
                             "dir_encoding_config": {"otype": "SphericalHarmonics", "degree": 4},
                             "mlp_network_config": {"otype": "VanillaMLP", "activation": "ReLU",
                                                    "output_activation": "none", "n_neurons": 64, "n_hidden_layers": 2},
                             "color_activation": "sigmoid"}})
    ncfg.pop("light")
    torch.manual_seed(6)
    neus = NeuSModel(mg.Cfg(ncfg))
    with torch.no_grad():
        neus.geometry.encoding.encoding.encoding.params.mul_(10.0)
        l0 = neus.geometry.network.layers[0]
        l0.weight_v[:, 3:] = torch.randn_like(l0.weight_v[:, 3:]) * 0.3
    from models import geometry as rgeo
    neus.geometry.contraction_type = rgeo.ContractionType.AABB      # (neus.py:57 leaves it to the caller)
    neus.eval()
    neus.update_step(0, 0)
    neus.background_color = torch.ones(3)
    with torch.no_grad():
        nout = neus.forward_(rays)
    call = neus.occupancy_grid.calls[0]
    print("neus samples", call[0].numel(), "valid rays", int((nout["opacity"][:, 0] > 0.5).sum()))
    nsaved = {"rays": rays, "shell": np.array(SHELL), "fd_eps": np.array(neus.geometry._finite_difference_eps),
              "render_step_size": np.array(neus.render_step_size),
              "primary_ri": call[0], "primary_ts": call[1], "primary_te": call[2]}
    for k, v in nout.items():
        if isinstance(v, torch.Tensor):
            nsaved["out__" + k] = v
    for k, v in neus.state_dict().items():
        nsaved["p__" + k] = v
    mg.save("models_neus.npz", **nsaved)

    # ---- the same at the sizes config[2..4] run: L = 16 levels, T = 2^19 entries, 2 x 128 SDF network
    # with 48 features, 128-wide radiance network -- the fused H = 128 / L = 16 kernels against the reference's own forward_
    # rather than only the H = 32 per-wave ones.  The 58 MB table is NOT stored: it is regenerated from BIG_TABLE_SEED
    # (big_table() below; tests/test_gpu_model_fixtures.py calls the same function).
    bcfg = model_cfg(False, False)
    bcfg.update({"name": "neus", "variance": {"init_val": 0.45, "modulate": False}, "num_samples_per_ray": 512,
                 "texture": {"name": "volume-radiance", "input_feature_dim": 48 + 3,
                             "dir_encoding_config": {"otype": "SphericalHarmonics", "degree": 4},
                             "mlp_network_config": {"otype": "VanillaMLP", "activation": "ReLU",
                                                    "output_activation": "none", "n_neurons": 128, "n_hidden_layers": 2},
                             "color_activation": "sigmoid"}})
    bcfg.pop("light")
    bcfg["geometry"] = dict(bcfg["geometry"], feature_dim=48)
    bcfg["geometry"]["xyz_encoding_config"] = dict(bcfg["geometry"]["xyz_encoding_config"], n_levels=16, log2_hashmap_size=19,
                                                   base_resolution=32, per_level_scale=1.447269237440378, start_level=16)
    bcfg["geometry"]["mlp_network_config"] = dict(bcfg["geometry"]["mlp_network_config"], n_neurons=128)
    torch.manual_seed(7)
    big = NeuSModel(mg.Cfg(bcfg))
    brays = camera_rays(16, 16, seed=4)
    with torch.no_grad():
        prm = big.geometry.encoding.encoding.encoding.params
        prm.copy_(big_table(prm.numel()))
        l0 = big.geometry.network.layers[0]
        l0.weight_v[:, 3:] = torch.randn_like(l0.weight_v[:, 3:]) * 0.3
    big.geometry.contraction_type = rgeo.ContractionType.AABB
    big.eval()
    big.update_step(0, 0)
    big.background_color = torch.ones(3)
    with torch.no_grad():
        bout = big.forward_(brays)
    call = big.occupancy_grid.calls[0]
    print("neus L16 H128 samples", call[0].numel(), "valid rays", int((bout["opacity"][:, 0] > 0.5).sum()))
    assert int((bout["opacity"][:, 0] > 0.5).sum()) > 20
    bsaved = {"rays": brays, "shell": np.array(SHELL), "fd_eps": np.array(big.geometry._finite_difference_eps),
              "render_step_size": np.array(big.render_step_size), "table_seed": np.array(BIG_TABLE_SEED),
              "n_table": np.array(prm.numel()), "primary_ri": call[0], "primary_ts": call[1], "primary_te": call[2]}
    for k, v in bout.items():
        if isinstance(v, torch.Tensor):
            bsaved["out__" + k] = v
    for k, v in big.state_dict().items():
        if not k.endswith("encoding.params"):
            bsaved["p__" + k] = v
    mg.save("models_neus_l16_h128.npz", **bsaved)

    pbr_l16_h128(SplitMixedOCCModel)


def pbr_l16_h128(SplitMixedOCCModel):
    # ---- the FULL PBR model at the sizes the shipped kernel family runs: SplitMixedOCCModel.forward_,
    # stage 1 with secondary-ray occlusion, L = 16 levels, T = 2^19 entries (table from its seed), 2 x 128 SDF network with
    # 48 features, 128-wide radiance networks (albedo / env / secondary 4 hidden layers, roughness / metallic 2), 256 rays.
    # The small fixtures above (L = 4, H = 32, 64-wide radiance networks) run the round-1 per-wave SDF kernels and per-layer
    # kernels of another width; this one runs the x2 SDF kernels at H = 128, the L = 16 stencil gather and the 128-wide layer
    # kernels against the reference's own forward_.
    pcfg = model_cfg(True, True, tex_hidden=128)
    pcfg["geometry"] = dict(pcfg["geometry"], feature_dim=48)
    pcfg["geometry"]["xyz_encoding_config"] = dict(pcfg["geometry"]["xyz_encoding_config"], n_levels=16, log2_hashmap_size=19,
                                                   base_resolution=32, per_level_scale=1.447269237440378, start_level=16)
    pcfg["geometry"]["mlp_network_config"] = dict(pcfg["geometry"]["mlp_network_config"], n_neurons=128)
    pcfg["texture"] = dict(pcfg["texture"], input_feature_dim=48)
    pcfg["variance"] = {"init_val": 0.45, "modulate": False}
    torch.manual_seed(8)
    pbr = SplitMixedOCCModel(mg.Cfg(pcfg))
    prays = camera_rays(16, 16, seed=4)
    with torch.no_grad():
        prm = pbr.geometry.encoding.encoding.encoding.params
        prm.copy_(big_table(prm.numel()))
        l0 = pbr.geometry.network.layers[0]
        l0.weight_v[:, 3:] = torch.randn_like(l0.weight_v[:, 3:]) * 0.3
        # (with the hash columns un-zeroed, weight_norm flattens the sphere init until the whole box is inside the surface:
        # shifted so that the rays cross sdf = 0 inside the occupied shell)
        pbr.geometry.network.layers[-1].bias[0] += 0.25
        pbr.texture.FG_LUT = otex.synthetic_fg_lut(LUT_RES)
        # (the environment map from a seed of its own instead of the constructor's draw: regenerated by the tests, not stored)
        pbr.emitter.base.copy_(big_envmap(pbr.emitter.base.shape))
    pbr.eval()
    pbr.update_step(0, 0)
    assert pbr.stage == 1
    pbr.background_color = torch.ones(3)
    with torch.no_grad():
        pbr.emitter.build_mips()
        pout = pbr.forward_(prays, relighting=False)
    calls = pbr.occupancy_grid.calls
    n_valid = int((pout["opacity"][:, 0] > 0.5).sum())
    print("split-mixed-occ L16 H128: samples", calls[0][0].numel(), "valid rays", n_valid, "secondary samples",
          calls[1][0].numel())
    assert n_valid > 30 and calls[1][0].numel() > 100
    psaved = {"rays": prays, "shell": np.array(SHELL), "lut_res": np.array(LUT_RES), "table_seed": np.array(BIG_TABLE_SEED),
              "n_table": np.array(prm.numel()), "envmap_seed": np.array(BIG_ENVMAP_SEED),
              "fd_eps": np.array(pbr.geometry._finite_difference_eps), "render_step_size": np.array(pbr.render_step_size),
              "primary_ri": calls[0][0], "primary_ts": calls[0][1], "primary_te": calls[0][2],
              "secondary_ri": calls[1][0], "secondary_ts": calls[1][1], "secondary_te": calls[1][2]}
    for k, v in pout.items():
        if isinstance(v, torch.Tensor) and v.dtype in (torch.float32, torch.bool, torch.int32) and not k.endswith("_bg"):
            psaved["out__" + k] = v
    for k, v in pbr.state_dict().items():
        if not (k.endswith("encoding.params") or k.endswith("FG_LUT") or k == "emitter.base"):
            psaved["p__" + k] = v
    mg.save("models_split_mixed_occ_l16_h128.npz", **psaved)


BIG_TABLE_SEED = 77
BIG_ENVMAP_SEED = 78


def big_envmap(shape, seed=BIG_ENVMAP_SEED):
    """The environment map of the large PBR fixture, from its seed (lib/pbr/light.py:139's rand * 0.5 + 0.25 draw)."""
    return torch.rand(tuple(shape), generator=torch.Generator().manual_seed(seed)) * 0.5 + 0.25


def big_table(n, seed=BIG_TABLE_SEED):
    """The L = 16, T = 2^19 hash table of the large fixture, from its seed (58 MB: not stored)."""
    return (torch.rand(n, generator=torch.Generator().manual_seed(seed)) * 2 - 1) * 1e-3


if __name__ == "__main__":
    main()
