"""CPU: how well conditioned the REFERENCE's own backward is in the late-training regime, measured on the oracle
("perturb its inputs by one ulp and see how far its gradients move; use that as the gate").

lib/nerfacc/cuda/csrc/render_weight.cu:139-151 forms d_alpha_j = (gw_j T_j - accum) / max(1 - alpha_j, 1e-10) with a running
total that is summed and then decremented; once a ray has saturated (alpha exactly 1 at the surface crossing, inv_s in the
thousands) ``accum`` is a rounding residue and the clamp multiplies it by up to 1e10.  The parity tests of that regime
(tests/test_gpu_late_regime.py) gate each gradient tensor at max(SURVEY 8(d)'s figure, 3 x the movement measured by
helpers.oracle_gradient_sensitivity in the case itself); this file pins the two facts that gate rests on, without a GPU:
at inv_s 1808 the oracle's gradients move by more than SURVEY's gate when its stencil inputs move by one fp32 ulp, and at
inv_s 403 (no exactly-saturated alphas) they do not."""
import pytest
import torch

import oracle
from helpers import camera_rays, late_regime_field, oracle_gradient_sensitivity


@pytest.mark.parametrize("variance,saturated", [(0.75, True), (0.6, False)])
def test_reference_backward_conditioning_in_the_late_regime(variance, saturated):
    meta, table, mlp = late_regime_field(hidden=64)
    var = torch.tensor(variance, requires_grad=True)
    rays = camera_rays(32, 32, seed=21)
    n = rays.shape[0]
    u = torch.rand(n, generator=torch.Generator().manual_seed(22))
    roi = torch.tensor([-1.5] * 3 + [1.5] * 3)
    step = 3.0 * 3 ** 0.5 / 256
    eps = oracle.progressive_fd_eps(1.5, 32, 1.447269237440378, 16)
    ci, cs, ce = oracle.ray_marching(rays[:, :3].contiguous(), rays[:, 3:].contiguous(), scene_aabb=roi, near_plane=0.0,
                                     far_plane=1e10, render_step_size=step, stratified_u=u)
    with torch.no_grad():
        r0 = oracle.neus_geometry_render(rays, ci, cs, ce, table, meta, mlp, var, radius=1.5, fd_eps=eps)
    keep = oracle.render_visibility(r0["alphas"], ray_indices=ci, n_rays=n, early_stop_eps=1e-4)
    ri, ts, te = ci[keep], cs[keep], ce[keep]
    sdf7, alphas = r0["sdf7"][keep], r0["alphas_own"][keep]
    n_sat = int((alphas == 1.0).sum())
    assert (n_sat > 50) == saturated, n_sat
    g = torch.Generator().manual_seed(23)
    cot = {"opacity": torch.randn(n, 1, generator=g), "depth": torch.randn(n, 1, generator=g)}
    leaves = {"table": table, "variance": var}
    for i, p in enumerate(mlp):
        for k in p:
            leaves[f"{i}.{k}"] = p[k]
    alpha_grads = []

    def render(s7):
        out = oracle.neus_geometry_render(rays, ri, ts, te, table, meta, mlp, var, radius=1.5, fd_eps=eps, sdf7_given=s7,
                                          alphas_given=alphas)
        out["alphas"].register_hook(lambda g_: alpha_grads.append(float(g_.abs().max())))
        return out

    base, moved = oracle_gradient_sensitivity(render, leaves, sdf7, cot, trials=2, seed=24)
    print(f"inv_s {float(torch.exp(torch.tensor(10 * variance))):.0f}: {ri.numel()} samples, {n_sat} alphas exactly 1, "
          f"|d_alpha| max {alpha_grads[0]:.3g}; oracle vs itself one ulp away: " + ", ".join(f"{k} {v:.1e}" for k, v in moved.items()))
    assert all(float(b.abs().max()) > 0 for b in base.values())
    if saturated:
        # the residue amplifier is active (the closed form bounds |d_alpha| by ~10) ...
        assert alpha_grads[0] > 1e3
        # ... and one ulp of the stencil moves EVERY gradient tensor by more than SURVEY 8(d)'s 1e-4
        assert min(moved.values()) > 1e-4 and max(moved.values()) > 5e-4, moved
    else:
        assert alpha_grads[0] < 100.0
        assert max(moved.values()) < 1e-4, moved
