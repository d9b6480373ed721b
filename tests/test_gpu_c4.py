"""config[4]-shaped end-to-end check: the full split-mixed-occ model at the yaml's sizes (128-wide SDF and texture
MLPs, 48 features, 16-level grid, 512^2 environment map, occupancy pruning, secondary rays, curvature, stage switch)
driven like systems/split_occ.py's training_step -- ray generation, update_step, build_mips, forward, loss tail,
backward, Adam -- for a few steps on synthetic images.  Checks that every piece composes at the real sizes (finite losses,
parameters move, stage switches).  Numerical parity of the training step itself -- HIP vs the oracle step by step, and the
convergence proxy for the PSNR gate -- is tests/test_gpu_convergence.py; the N-rank step is tests/test_gpu_dist_step.py."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def c4_config(n_levels=16, log2_T=19):
    """The yaml's model node (rise_sdf_amd.config.tensoir_model_config) with the stage switch pulled to step 2."""
    from rise_sdf_amd.config import tensoir_model_config
    return tensoir_model_config(n_levels=n_levels, log2_T=log2_T, split_sum_kick_in_step=2)


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
