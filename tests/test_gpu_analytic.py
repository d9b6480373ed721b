"""GPU parity for the analytic-normal rows (H4 analytic variant, H5 curvature): the hash grid's input-gradient
kernels and the composed field gradient vs the twice-differentiable fp64 oracle (oracle/analytic.py)."""
import pytest
import torch

import oracle
from oracle import analytic as A
from helpers import rel_err
from test_gpu_model import model_config, oracle_params

pytestmark = pytest.mark.gpu


def _meta_pair(n_levels, log2_T, F=2):
    from rise_sdf_amd import tinycudann as tcnn
    enc = tcnn.Encoding(3, {"otype": "HashGrid", "n_levels": n_levels, "n_features_per_level": F,
                            "log2_hashmap_size": log2_T, "base_resolution": 16, "per_level_scale": 1.5})
    meta_o, n = oracle.grid_meta(n_levels, F, log2_T, 16, 1.5)
    assert n == enc.params.numel()
    return enc, meta_o


@pytest.mark.parametrize("n_levels,n_active,F", [(6, 6, 2), (6, 4, 2), (4, 4, 4), (5, 5, 1)])
def test_hashgrid_dx_and_double_backward(dev, n_levels, n_active, F):
    from rise_sdf_amd import ops
    enc, meta_o = _meta_pair(n_levels, 12, F)
    g = torch.Generator().manual_seed(n_levels * 10 + F)
    table = (torch.rand(enc.params.numel(), generator=g) * 2 - 1) * 0.1
    S, LF = 3000, n_levels * F
    x = torch.rand(S, 3, generator=g)
    dy = torch.randn(S, 3 + LF, generator=g)      # leading 3 columns = xyz pass-through slots (ignored)
    gdx = torch.randn(S, 3, generator=g)
    xg, tg, dyg = x.to(dev).requires_grad_(True), table.to(dev).requires_grad_(True), dy.to(dev).requires_grad_(True)
    dx = ops.hashgrid_dx(xg, tg, dyg, enc.meta, n_active, 3)
    gx, gt, gdy = torch.autograd.grad(dx, [xg, tg, dyg], gdx.to(dev))
    # oracle: dx = d <enc(x), dy> / dx, then its gradients
    x64, t64, d64 = x.double().requires_grad_(True), table.double().requires_grad_(True), dy.double().requires_grad_(True)
    e = A.hashgrid_encode_t(x64, t64, meta_o, n_active)
    (dx_o,) = torch.autograd.grad((e * d64[:, 3:]).sum(), x64, create_graph=True)
    gx_o, gt_o, gdy_o = torch.autograd.grad(dx_o, [x64, t64, d64], gdx.double())
    assert rel_err(dx, dx_o) < 1e-5
    assert rel_err(gdy, gdy_o) < 1e-5
    assert float(gdy[:, :3].abs().max()) == 0.0
    assert rel_err(gt, gt_o) < 1e-4
    assert rel_err(gx, gx_o) < 1e-4
    # the encode op's own d/dx (first-order route through autograd) is the same kernel
    xg2 = x.to(dev).requires_grad_(True)
    out = ops.hashgrid_encode(xg2, tg, enc.meta, n_active_levels=n_active, include_xyz=True)
    (dx2,) = torch.autograd.grad(out, xg2, dy.to(dev))
    assert rel_err(dx2, dx_o + 2.0 * dy[:, :3].double()) < 1e-5


def _analytic_model(dev, hidden=32, n_levels=4):
    import rise_sdf_amd as R
    torch.manual_seed(0)
    cfg = model_config(n_levels=n_levels, hidden=hidden)
    model = R.make("neus", cfg).to(dev)
    model.train()
    with torch.no_grad():
        model.geometry.encoding.encoding.encoding.params.mul_(300.0)
        l0 = model.geometry.network.layers[0]
        l0.weight_v[:, 3:] = torch.randn_like(l0.weight_v[:, 3:]) * 0.3
    model.geometry.update_step(0, 0)
    return model


def test_analytic_gradient_and_curvature_match_oracle(dev):
    model = _analytic_model(dev)
    geo = model.geometry
    meta, table, mlp, _ = oracle_params(model)
    t64 = table.detach().double().requires_grad_(True)
    mlp64 = [{k: v.detach().double().requires_grad_(True) for k, v in p.items()} for p in mlp]
    g = torch.Generator().manual_seed(4)
    pts = (torch.rand(1500, 3, generator=g) * 2 - 1) * 1.3
    rnd = torch.rand(1500, 3, generator=g)
    # --- analytic field gradient (grad_type 'analytic')
    out, grad = geo.field_with_analytic_grad(pts.to(dev))
    sdf_o, grad_o, feat_o = A.volume_sdf_analytic(pts.double(), t64, meta, mlp64, radius=1.5)
    assert rel_err(out, feat_o) < 1e-5
    assert float((grad.detach().cpu().double() - grad_o.detach()).abs().max()) < 1e-4 * float(grad_o.abs().max())
    gg = torch.randn(1500, 3, generator=g)
    eik_g = ((grad * gg.to(dev)).sum())
    eik_o = ((grad_o * gg.double()).sum())
    eik_g.backward()
    eik_o.backward()
    gt = geo.encoding.encoding.encoding.params.grad.cpu()
    assert rel_err(gt, t64.grad) < 1e-3
    lin = [m for m in geo.network.layers if isinstance(m, torch.nn.Linear)]
    for m, p in zip(lin, mlp64):
        assert rel_err(m.weight_v.grad, p["v"].grad) < 1e-3
        assert rel_err(m.weight_g.grad, p["g"].grad) < 1e-3
    # --- curvature (FD normals from the oracle's FD field so that both sides start from the same grad)
    model.zero_grad(set_to_none=True)
    t64.grad = None
    for p in mlp64:
        for v in p.values():
            v.grad = None
    eps = geo._finite_difference_eps
    sdf_g, grad_fd_g, _ = geo(pts.to(dev), with_grad=True, with_feature=True)
    lap_g = geo.curvature(pts.to(dev), grad_fd_g, rnd.to(dev))
    offs = torch.tensor([[eps, 0, 0], [-eps, 0, 0], [0, eps, 0], [0, -eps, 0], [0, 0, eps], [0, 0, -eps]],
                        dtype=torch.float64)
    pd = (pts.double()[:, None, :] + offs).clamp(-1.5, 1.5)
    sd = A.field(pd.reshape(-1, 3), t64, meta, mlp64, 1.5)[:, 0].view(-1, 6)
    grad_fd_o = 0.5 * (sd[:, 0::2] - sd[:, 1::2]) / eps
    lap_o = A.curvature(pts.double(), grad_fd_o, rnd.double(), t64, meta, mlp64, radius=1.5)
    # acos amplifies fp32 noise near 0 / 1: absolute tolerance on the angle fraction
    assert float((lap_g.detach().cpu().double() - lap_o.detach()).abs().max()) < 2e-3
    lap_g.mean().backward()
    lap_o.mean().backward()
    gt = geo.encoding.encoding.encoding.params.grad.cpu()
    assert torch.nn.functional.cosine_similarity(gt[None].double(), t64.grad[None]).item() > 0.999
    assert rel_err(gt, t64.grad) < 5e-2


def test_volume_sdf_analytic_grad_type(dev):
    """grad_type='analytic' (configs/neus-*.yaml): forward returns the analytic gradient, eval mode too."""
    import rise_sdf_amd as R
    cfg = model_config(n_levels=4, hidden=32)
    cfg["geometry"]["grad_type"] = "analytic"
    torch.manual_seed(0)
    model = R.make("neus", cfg).to(dev)
    model.geometry.update_step(0, 0)
    g = torch.Generator().manual_seed(1)
    pts = (torch.rand(64, 3, generator=g) * 2 - 1).to(dev)
    model.eval()
    sdf, grad, feat = model.geometry(pts, with_grad=True, with_feature=True)
    assert not grad.requires_grad and not sdf.requires_grad
    # central differences of the same field (sphere init, smooth at this scale)
    h = 1e-3
    eye = torch.eye(3, device=dev)
    num = torch.stack([(model.geometry(pts + h * eye[d], with_grad=False, with_feature=False)
                        - model.geometry(pts - h * eye[d], with_grad=False, with_feature=False)) / (2 * h)
                       for d in range(3)], -1)
    assert float((grad - num).abs().max()) < 5e-3
    assert float(torch.nn.functional.cosine_similarity(grad, num, dim=-1).min()) > 0.9999
    # training mode: the gradient carries a graph to the parameters (eikonal loss)
    model.train()
    sdf, grad, feat = model.geometry(pts, with_grad=True, with_feature=True)
    ((grad.norm(dim=-1) - 1.0) ** 2).mean().backward()
    assert model.geometry.network.layers[0].weight_v.grad is not None


def test_softplus100_with_slope_is_torchs_pair(dev):
    """rise_sdf_amd/geometry.py's analytic sweep takes (softplus(z, beta=100), sigmoid(100 z)) from one kernel each way:
    values equal torch's two ops to an ulp (same libm calls), gradients of both outputs to 1e-6 -- over the threshold at
    100 z = 20, the saturated tails and zero."""
    import torch.nn.functional as F
    from rise_sdf_amd.texture_ops import softplus100_slope
    g = torch.Generator().manual_seed(0)
    z = torch.cat([torch.randn(4099, generator=g) * 0.05, torch.tensor([0.0, 0.2, 0.2000001, 0.3, -0.3, 5.0, -5.0, 1e-8])])
    z = z.to(dev).reshape(-1, 1).repeat(1, 3).contiguous()
    gh, gs = torch.randn(z.shape, generator=g).to(dev), torch.randn(z.shape, generator=g).to(dev)
    za = z.clone().requires_grad_(True)
    h, s = softplus100_slope(za)
    (h * gh + s * gs).sum().backward()
    zb = z.clone().requires_grad_(True)
    h_t, s_t = F.softplus(zb, beta=100), torch.sigmoid(100.0 * zb)
    (h_t * gh + s_t * gs).sum().backward()
    assert float((h - h_t).abs().max()) <= 2e-7 * float(h_t.abs().max())
    assert float((s - s_t).abs().max()) <= 2e-7
    assert float((za.grad - zb.grad).abs().max()) <= 2e-6 * float(zb.grad.abs().max())
    # one of the two outputs unused downstream
    zc = z.clone().requires_grad_(True)
    (softplus100_slope(zc)[0] * gh).sum().backward()
    zd = z.clone().requires_grad_(True)
    (F.softplus(zd, beta=100) * gh).sum().backward()
    assert float((zc.grad - zd.grad).abs().max()) <= 2e-6 * float(zd.grad.abs().max())
