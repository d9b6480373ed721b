"""config[4]-shaped end-to-end check: the full split-mixed-occ model at the yaml's sizes (128-wide SDF and texture
MLPs, 48 features, 16-level grid, 512^2 environment map, occupancy pruning, secondary rays, curvature, stage switch)
driven like systems/split_occ.py's training_step -- ray generation, update_step, build_mips, forward, loss tail,
backward, Adam -- for a few steps on synthetic images.  Checks that every piece composes and trains (finite losses,
parameters move, stage switches), not numerical parity (the pieces have their own parity tests)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def c4_config(n_levels=16, log2_T=19):
    from rise_sdf_amd import Config
    mlp = lambda n: {"otype": "VanillaMLP", "activation": "ReLU", "output_activation": "none", "n_neurons": 128,
                     "n_hidden_layers": n}
    return Config({
        "name": "split-mixed-occ", "indirect_pred": True, "relighting_threshold": 0.3, "radius": 1.5,
        "num_samples_per_ray": 1024, "num_samples_per_secondary_ray": 96, "train_num_rays": 256,
        "grid_prune": True, "grid_prune_occ_thre": 0.001, "randomized": True, "ray_chunk": 4096,
        "cos_anneal_end": 10000, "learned_background": False, "split_sum_kick_in_step": 2,
        "variance": {"init_val": 0.3, "modulate": False},
        "geometry": {
            "name": "volume-sdf", "radius": 1.5, "feature_dim": 48, "grad_type": "finite_difference",
            "finite_difference_eps": "progressive",
            "xyz_encoding_config": {"otype": "ProgressiveBandHashGrid", "n_levels": n_levels, "start_level": 6,
                                    "start_step": 6000, "update_steps": 500, "n_features_per_level": 2,
                                    "log2_hashmap_size": log2_T, "base_resolution": 32,
                                    "per_level_scale": 1.447269237440378, "include_xyz": True},
            "mlp_network_config": {"otype": "VanillaMLP", "activation": "ReLU", "output_activation": "none",
                                   "n_neurons": 128, "n_hidden_layers": 2, "sphere_init": True,
                                   "sphere_init_radius": 0.5, "weight_norm": True}},
        "texture": {"name": "volume-mixed-mip-split-occ", "input_feature_dim": 48, "other_dim": 3, "sample_size": 8,
                    "dir_encoding_config": {"otype": "SphericalHarmonics", "degree": 5, "reflected": True},
                    "metallic_mlp_network_config": mlp(2), "albedo_mlp_network_config": mlp(4),
                    "spec_mlp_network_config": mlp(4), "roughness_mlp_network_config": mlp(2),
                    "secondary_mlp_network_config": mlp(4),
                    "xyz_encoding_config": {"otype": "VanillaFrequency", "n_frequencies": 6},
                    "color_activation": "sigmoid"},
        "light": {"name": "envlight-mip-cube",
                  "envlight_config": {"hdr_filepath": None, "clamp": True, "nmf_format": False, "scale": 0.5,
                                      "bias": 0.25, "base_res": 512}},
    })


def test_full_model_trains_a_few_steps(dev):
    import rise_sdf_amd as R
    from rise_sdf_amd import ops
    from rise_sdf_amd.loss import loss_tail
    from rise_sdf_amd.ray_utils import get_ray_directions
    torch.manual_seed(0)
    model = R.make("split-mixed-occ", c4_config()).to(dev)
    model.train()
    lambdas = {"lambda_rgb_mse": 10.0, "lambda_rgb_phys_mse": 10.0, "lambda_mask": 0.1, "lambda_eikonal": 0.05,
               "lambda_sparsity": 0.01, "lambda_curvature": 1.0, "lambda_opaque": 0.0}
    opt = torch.optim.Adam([{"params": model.geometry.parameters(), "lr": 5e-3},
                            {"params": model.texture.parameters(), "lr": 5e-3},
                            {"params": model.variance.parameters(), "lr": 1e-3},
                            {"params": model.emitter.parameters(), "lr": 1e-2}], betas=(0.9, 0.999), eps=1e-12)
    # a resident synthetic dataset: 3 views of a grey disc on white (systems/split_occ.py keeps images on the GPU)
    V, H, W = 3, 64, 64
    g = torch.Generator().manual_seed(1)
    dirs = get_ray_directions(W, H, 70.0, 70.0, W / 2, H / 2, device=dev)
    c2w = torch.tensor([[[1.0, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]],
                        [[0.0, 0, 1, 4.0], [0, 1, 0, 0], [-1, 0, 0, 0]],
                        [[1.0, 0, 0, 0], [0, 0, 1, 4.0], [0, -1, 0, 0]]], device=dev)
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    disc = (((yy - H / 2) ** 2 + (xx - W / 2) ** 2) < (0.12 * W * 4) ** 2 / 16).float()
    masks = disc[None].expand(V, H, W).contiguous().to(dev)
    images = (0.5 * masks[..., None]).expand(V, H, W, 3).contiguous()
    p0 = model.geometry.encoding.encoding.encoding.params.detach().clone()
    base0 = model.emitter.base.detach().clone()
    stages = []
    for step in range(4):
        model.update_step(0, step)                       # occupancy update on step 0, stage switch at step 2
        stages.append(model.stage)
        n = 256
        index = torch.randint(0, V, (n,), generator=g).to(dev)
        y, x = torch.randint(0, H, (n,), generator=g).to(dev), torch.randint(0, W, (n,), generator=g).to(dev)
        model.background_color = torch.ones(3, device=dev)
        rays, rgb, fg = ops.gen_rays(index, y, x, dirs, c2w, images, masks, model.background_color, apply_mask=True)
        if model.stage:
            model.emitter.build_mips()
        out = model(rays)
        loss, terms = loss_tail(out, {"rgb": rgb, "fg_mask": fg}, lambdas, sparsity_scale=1.0)
        for name, value in model.regularizations(out).items():
            loss = loss + 0.05 * value
        assert math.isfinite(float(loss)), (step, {k: float(v) for k, v in terms.items()})
        assert "sdf_laplace_samples" in out and out["sdf_laplace_samples"].shape == out["sdf_samples"].shape
        if model.stage:
            assert out["comp_rgb_phys_full"].shape == (n, 3)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    assert stages == [0, 0, 1, 1]
    assert int(model.occupancy_grid.binaries.sum()) > 0
    assert float((model.geometry.encoding.encoding.encoding.params.detach() - p0).abs().max()) > 0
    assert float((model.emitter.base.detach() - base0).abs().max()) > 0
    for prm in model.parameters():
        assert bool(torch.isfinite(prm).all())
