"""GPU edge cases: empty and tiny batches, odd widths, through the same entry points as the parity tests."""
import math

import pytest
import torch

import oracle
from helpers import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [0, 1, 31, 33])
@pytest.mark.parametrize("K,N,act", [(73, 128, "relu"), (84, 6, "sigmoid"), (5, 1, "none")])
def test_linear_small_batches_and_odd_widths(dev, n, K, N, act):
    from rise_sdf_amd import ops
    g = torch.Generator().manual_seed(n * 100 + K)
    x, w, b = torch.randn(n, K, generator=g), torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g)
    gy = torch.randn(n, N, generator=g)
    f = {"relu": torch.relu, "sigmoid": torch.sigmoid, "none": lambda t: t}[act]
    xo, wo, bo = [t.double().requires_grad_(True) for t in (x, w, b)]
    yo = f(torch.nn.functional.linear(xo, wo, bo))
    (yo * gy.double()).sum().backward()
    xg, wg, bg = [t.to(dev).requires_grad_(True) for t in (x, w, b)]
    yg = ops.linear(xg, wg, bg, act=act)
    assert yg.shape == (n, N)
    (yg * gy.to(dev)).sum().backward()
    if n == 0:
        assert float(wg.grad.abs().max()) == 0.0 and float(bg.grad.abs().max()) == 0.0
        return
    assert rel_err(yg, yo) < 1e-5
    assert rel_err(xg.grad, xo.grad) < 1e-5
    assert rel_err(wg.grad, wo.grad) < 2e-5
    assert rel_err(bg.grad, bo.grad) < 2e-5


def test_empty_batches_of_the_frontend_and_loss(dev):
    from rise_sdf_amd import ops
    from rise_sdf_amd.loss import loss_tail
    e64 = torch.zeros(0, dtype=torch.int64, device=dev)
    rays, rgb, fg = ops.gen_rays(torch.zeros(1, dtype=torch.int64, device=dev), e64, e64,
                                 torch.zeros(4, 4, 3, device=dev), torch.zeros(1, 3, 4, device=dev),
                                 torch.zeros(1, 4, 4, 3, device=dev), torch.zeros(1, 4, 4, device=dev),
                                 torch.ones(3, device=dev), apply_mask=True)
    assert rays.shape == (0, 6) and rgb.shape == (0, 3) and fg.shape == (0,)
    # occupancy update with no touched cells only re-thresholds
    occs = torch.tensor([0.0, 0.2, 0.4, 0.6], device=dev)
    binary = torch.zeros(4, dtype=torch.uint8, device=dev)
    ops.occ_update(occs, binary, e64, torch.zeros(0, device=dev), 0.95, 0.25)
    assert binary.cpu().tolist() == [0, 0, 1, 1]
    # a batch whose rays are all invalid and that has no samples: means over nothing are NaN and, as in the
    # reference (NaN * lambda), so is the weighted sum
    N = 8
    out = {"comp_rgb_full": torch.rand(N, 3, device=dev, requires_grad=True),
           "opacity": torch.rand(N, 1, device=dev).clamp(0.1, 0.9).requires_grad_(True),
           "rays_valid_full": torch.zeros(N, 1, dtype=torch.bool, device=dev),
           "sdf_samples": torch.zeros(0, device=dev, requires_grad=True),
           "sdf_grad_samples": torch.zeros(0, 3, device=dev, requires_grad=True)}
    batch = {"rgb": torch.rand(N, 3, device=dev), "fg_mask": torch.ones(N, device=dev)}
    loss, terms = loss_tail(out, batch, {"lambda_mask": 1.0, "lambda_opaque": 0.1})
    assert math.isnan(float(terms["loss_rgb_mse"])) and math.isnan(float(terms["loss_eikonal"]))
    assert math.isfinite(float(terms["loss_mask"])) and math.isfinite(float(terms["loss_opaque"]))
    assert math.isnan(float(loss))
    loss.backward()   # must not fault


@pytest.mark.parametrize("S", [1, 7, 33])
def test_fused_field_tiny_sample_counts(dev, S):
    """The fused stencil kernels on fewer samples than one 32-row tile."""
    import oracle
    from rise_sdf_amd import ops
    from test_gpu_model import model_config, oracle_params
    import rise_sdf_amd as R
    torch.manual_seed(0)
    model = R.make("neus", model_config(n_levels=4, hidden=32)).to(dev)
    model.train()
    model.geometry.update_step(0, 0)
    g = torch.Generator().manual_seed(S)
    n_rays = 3
    rays_o = torch.randn(n_rays, 3, generator=g) * 0.2
    rays_d = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=g), dim=-1)
    ri = torch.sort(torch.randint(0, n_rays, (S,), generator=g)).values
    ts = torch.rand(S, generator=g)
    te = ts + 0.01
    geo = model.geometry
    sdf7t, feat = geo.sdf7_from_rays(rays_o.to(dev), rays_d.to(dev), ri.to(dev), ts.to(dev), te.to(dev), want_feature=True)
    assert sdf7t.shape == (7, S) and feat.shape == (S, 13)
    out7 = geo.field7_from_rays(rays_o.to(dev), rays_d.to(dev), ri.to(dev), ts.to(dev), te.to(dev))
    ref = out7.view(S, 7, -1)
    assert torch.allclose(sdf7t.t(), ref[:, :, 0], rtol=1e-5, atol=1e-6)
    assert torch.allclose(feat, ref[:, 0], rtol=1e-5, atol=1e-6)
    (sdf7t.sum() + feat.sum()).backward()
    assert bool(torch.isfinite(geo.encoding.encoding.encoding.params.grad).all())


def test_occ_eval_alpha_kernel(dev):
    """A2 (models/split_mixed_occ.py:108-119): the occupancy-update alpha as one kernel vs the oracle's formula."""
    from rise_sdf_amd import ops
    g = torch.Generator().manual_seed(0)
    sdf = (torch.rand(5000, generator=g) * 2 - 1) * 0.05
    sdf[:4] = torch.tensor([0.0, 1.0, -1.0, 1e-4])
    for v in (0.3, 0.05, 0.9):
        var = torch.tensor(v)
        got = ops.occ_alpha(sdf.to(dev), var.to(dev), 0.00507421875)
        want = oracle.occ_alpha(sdf, oracle.inv_s_from_variance(var), 0.00507421875)
        assert got.shape == (5000, 1)
        assert torch.allclose(got.cpu().view(-1), want.view(-1), rtol=1e-5, atol=1e-6), v


def test_fused_mlp_is_fp32_accurate(dev):
    """The split-bf16 fused kernels against an fp64 evaluation of the same weights on the same hash features, next to
    plain fp32 torch: the fused SDF must be at least as close to fp64 as torch's fp32 GEMM chain, and the error of the
    finite-difference normal must be of the size an fp32 ulp of SDF noise explains (|err| ~ ulp / eps), which is the
    justification of the gradient tolerances in test_gpu_model.py."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import rise_sdf_amd as R
    from rise_sdf_amd import fused, ops
    for hidden in (64, 128):
        torch.manual_seed(0)
        model = R.make("neus", bench.c1_config(hidden=hidden)).to(dev)
        geo = model.geometry
        with torch.no_grad():
            geo.encoding.encoding.encoding.params.uniform_(-1e-1, 1e-1)
            l0 = geo.network.layers[0]
            l0.weight_v[:, 3:] = torch.randn_like(l0.weight_v[:, 3:]) * 0.3
        model.train()
        geo.update_step(0, 20000)
        S = 60000
        g = torch.Generator().manual_seed(1)
        pts = (torch.rand(S, 3, generator=g) * 2 - 1) * 1.4
        x7 = ops.fd_taps(pts.to(dev), geo.radius, geo._finite_difference_eps)
        enc = geo.encoding(x7.view(-1, 3), fd7_eps_unit=geo._eps_unit()).detach()       # bit-exact gather
        wb = [(w.detach(), b.detach()) for w, b in geo.network.effective_weights()]

        def mlp(x, dtype):
            h = x.to(dtype)
            for i, (w, b) in enumerate(wb):
                h = torch.nn.functional.linear(h, w.to(dtype), b.to(dtype))
                if i < len(wb) - 1:
                    h = torch.nn.functional.softplus(h, beta=100)
            return h[:, 0]
        ref = mlp(enc, torch.float64)
        t32 = mlp(enc, torch.float32).double()
        grid, n_active = geo.encoding._hash()
        sdf7t, _ = fused.sdf_field_fd7(x7.permute(1, 0, 2).contiguous(), grid.params, geo.network.effective_weights(),
                                       grid.meta, grid.n_levels if n_active is None else n_active,
                                       geo.encoding.xyz_scale, geo.encoding.xyz_offset, geo._eps_unit(), False)
        fs = sdf7t.detach().t().reshape(-1).double()
        e_fused, e_t32 = float((fs - ref).abs().max()), float((t32 - ref).abs().max())
        assert e_fused < 4e-7 and e_fused <= 1.5 * e_t32 + 1e-7, (hidden, e_fused, e_t32)
        eps = geo._finite_difference_eps
        fd = lambda s: 0.5 * (s.view(-1, 7)[:, 1::2] - s.view(-1, 7)[:, 2::2]) / eps   # noqa: E731
        e_n = float((fd(fs) - fd(ref)).abs().max())
        assert e_n < 2.0 * e_fused / eps + 1e-6 and e_n < 2e-3, (hidden, e_n, e_fused / eps)


def test_transient_workspaces(dev):
    """_lib.workspace: one growing arena per (tag, device, stream) for buffers produced and consumed inside one backward
    call (the fused field's gradient planes, record queues, d(h2)): a second request of the same or a smaller size returns
    the same memory, a larger one a new buffer, another stream its own; tcnn.free_temporary_memory() of the drop-in
    releases them; the fused field's table gradient does not depend on what an earlier, larger call left in the arena."""
    from rise_sdf_amd import _lib, tinycudann
    _lib.free_workspaces()
    a = _lib.workspace("t.test", 1 << 20, dev)
    assert a.dtype == torch.uint8 and a.numel() == 1 << 20 and a.data_ptr() % 256 == 0
    b = _lib.workspace("t.test", 1 << 19, dev)
    assert b.data_ptr() == a.data_ptr() and b.numel() == 1 << 19
    f = _lib.workspace_f32("t.test", (16, 7, 100, 2), dev)
    assert f.dtype == torch.float32 and f.shape == (16, 7, 100, 2) and f.data_ptr() == a.data_ptr()
    c = _lib.workspace("t.test", 1 << 24, dev)
    assert c.numel() == 1 << 24
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        d = _lib.workspace("t.test", 1 << 20, dev)
    assert d.data_ptr() != c.data_ptr()
    assert len([k for k in _lib._WORKSPACES if k[0] == "t.test"]) == 2
    tinycudann.free_temporary_memory()
    assert not _lib._WORKSPACES

    # a backward that finds a larger call's leftovers in the arena gives what a fresh arena gives
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from helpers import camera_rays
    import argparse
    model = bench.build_model(dev, argparse.Namespace(hidden=64))     # (non-zero hash-feature columns in the first layer)
    rays = camera_rays(48, 48, seed=3).to(dev)
    u = torch.rand(rays.shape[0], generator=torch.Generator().manual_seed(5)).to(dev)

    def table_grad(n):
        for p in model.parameters():
            p.grad = None
        s0 = (rays.shape[0] - n) // 2                        # (the middle of the view: rays that meet the surface)
        out = model.forward_(rays[s0:s0 + n], stratified_u=u[s0:s0 + n])
        (out["opacity"].sum() + out["depth"].sum() + out["comp_normal_raw"].sum()).backward()
        return model.geometry.encoding.encoding.encoding.params.grad.clone()

    table_grad(2304)                      # the larger call: grows the arenas
    assert any(k[0].startswith("fd7.") for k in _lib._WORKSPACES)
    reused = table_grad(500)
    _lib.free_workspaces()
    fresh = table_grad(500)
    scale = float(fresh.abs().max())
    assert scale > 0 and float((reused - fresh).abs().max()) <= 2e-6 * scale      # (fp32 flush order only)
