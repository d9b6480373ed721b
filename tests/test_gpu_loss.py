"""GPU parity for the loss tail (N1): fused reductions vs the oracle's restatement of systems/split_occ.py:163-215."""
import pytest
import torch

import oracle
from helpers import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("stage,curv", [(0, False), (1, True)])
def test_loss_tail_matches_oracle(dev, stage, curv):
    from rise_sdf_amd.loss import loss_tail
    g = torch.Generator().manual_seed(stage)
    N, S = 1777, 40001
    out = {"comp_rgb_full": torch.rand(N, 3, generator=g), "opacity": torch.rand(N, 1, generator=g) * 1.2 - 0.1,
           "rays_valid_full": torch.rand(N, 1, generator=g) > 0.3, "sdf_samples": torch.randn(S, generator=g) * 0.3,
           "sdf_grad_samples": torch.randn(S, 3, generator=g) * 0.7}
    out["sdf_grad_samples"][:3] = 0.0
    out["opacity"][:4, 0] = torch.tensor([0.0, 1.0, 2e-3, 1 - 2e-3])   # outside / just inside the clamp
    if stage:
        out["comp_rgb_phys_full"] = torch.rand(N, 3, generator=g)
    if curv:
        out["sdf_laplace_samples"] = torch.rand(S, generator=g) - 0.2
    batch = {"rgb": torch.rand(N, 3, generator=g), "fg_mask": (torch.rand(N, generator=g) > 0.5).float()}
    lambdas = {"lambda_rgb_mse": 10.0, "lambda_rgb_l1": 0.3, "lambda_rgb_phys_mse": 5.0, "lambda_rgb_phys_l1": 0.2,
               "lambda_mask": 0.1, "lambda_opaque": 0.05, "lambda_eikonal": 0.1, "lambda_sparsity": 0.01,
               "lambda_curvature": 1.0 if curv else 0.0}
    diff = ["comp_rgb_full", "opacity", "sdf_samples", "sdf_grad_samples"] + (["comp_rgb_phys_full"] if stage else []) \
        + (["sdf_laplace_samples"] if curv else [])
    o64 = {k: (v.double().requires_grad_(True) if k in diff else v) for k, v in out.items()}
    b64 = {k: v.double() for k, v in batch.items()}
    loss_o, terms_o = oracle.loss_tail(o64, b64, lambdas, sparsity_scale=20.0, stage=stage)
    grads_o = torch.autograd.grad(loss_o, [o64[k] for k in diff])
    og = {k: (v.to(dev).requires_grad_(True) if k in diff else v.to(dev)) for k, v in out.items()}
    loss_g, terms_g = loss_tail(og, {k: v.to(dev) for k, v in batch.items()}, lambdas, sparsity_scale=20.0)
    assert abs(float(loss_g) - float(loss_o)) < 2e-6 * abs(float(loss_o))
    for k, v in terms_o.items():
        assert abs(float(terms_g["loss_" + k]) - float(v)) <= 2e-6 * abs(float(v)) + 1e-9, k
    grads_g = torch.autograd.grad(loss_g, [og[k] for k in diff])
    for k, a, b in zip(diff, grads_g, grads_o):
        assert rel_err(a, b) < 2e-5, k
