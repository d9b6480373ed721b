"""The registered 'neus' model outside the fused SDF-only case (round-1 advisor findings):
  * with a radiance network (models/neus.py:240-317: comp_rgb + the _bg / _full dictionaries),
  * with grad_type 'analytic' (configs/neus-blender.yaml) -- normals must come from the analytic path, not from FD,
  * the secondary-ray feature query propagates the full d/d(position) (models/split_mixed_occ.py:315)."""
import pytest
import torch
import torch.nn.functional as F

import oracle
from oracle import analytic as OA
from oracle import texture as OT
from helpers import camera_rays, rel_err
from test_gpu_model import model_config, oracle_params

pytestmark = pytest.mark.gpu
ROI = torch.tensor([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5])


def _prep(model, dev):
    with torch.no_grad():
        model.geometry.encoding.encoding.encoding.params.mul_(300.0)
        l0 = model.geometry.network.layers[0]
        l0.weight_v[:, 3:] = torch.randn_like(l0.weight_v[:, 3:]) * 0.3
    model.geometry.update_step(0, 0)
    model.cos_anneal_ratio = 1.0


def test_neus_with_radiance_network(dev):
    import rise_sdf_amd as R
    torch.manual_seed(0)
    cfg = model_config(hidden=32, n_levels=4, feat=13)
    cfg["texture"] = {"name": "volume-radiance", "input_feature_dim": 13 + 3,
                      "dir_encoding_config": {"otype": "SphericalHarmonics", "degree": 4},
                      "mlp_network_config": {"otype": "VanillaMLP", "activation": "ReLU", "output_activation": "none",
                                             "n_neurons": 64, "n_hidden_layers": 2},
                      "color_activation": "sigmoid"}
    model = R.make("neus", R.Config(dict(cfg))).to(dev)
    model.train()
    _prep(model, dev)
    assert not model._fused_ok()
    rays = camera_rays(20, 20, seed=4)
    u = torch.rand(rays.shape[0], generator=torch.Generator().manual_seed(5))
    out = model.forward_(rays.to(dev), stratified_u=u.to(dev))
    for k in ("comp_rgb", "comp_rgb_bg", "comp_rgb_full", "num_samples_full", "rays_valid_full", "opacity", "depth"):
        assert k in out, k
    ri, ts, te = oracle.ray_marching(rays[:, :3].contiguous(), rays[:, 3:].contiguous(), scene_aabb=ROI,
                                     near_plane=0.0, far_plane=1e10, render_step_size=model.render_step_size,
                                     stratified_u=u)
    assert torch.equal(out["ray_indices"].cpu(), ri)
    meta, table, mlp, var = oracle_params(model)
    ref = oracle.neus_geometry_render(rays, ri, ts, te, table, meta, mlp, var, radius=1.5,
                                      fd_eps=model.geometry._finite_difference_eps)
    tex = [{"w": m.weight.detach().cpu().clone().requires_grad_(True), "b": m.bias.detach().cpu().clone().requires_grad_(True)}
           for m in model.texture.network.layers if isinstance(m, torch.nn.Linear)]
    t_dirs = rays[:, 3:][ri]
    # models/neus.py:258 normalises without the 1e-6 eps of split_mixed_occ; identical for |grad| ~ 1
    normal = F.normalize(ref["sdf_grad"], p=2, dim=-1)
    rgb = torch.sigmoid(OT.relu_mlp(torch.cat([ref["feature"], OT.sh_encode((t_dirs + 1) / 2, 4), normal], -1), tex))
    comp = oracle.accumulate_along_rays(ref["weights"], rgb, ray_indices=ri, n_rays=rays.shape[0])
    full = comp + 1.0 * (1.0 - ref["opacity"])
    assert torch.allclose(out["comp_rgb"].cpu(), comp, rtol=1e-4, atol=2e-5)
    assert torch.allclose(out["comp_rgb_full"].cpu(), full, rtol=1e-4, atol=2e-5)
    g = torch.randn(comp.shape, generator=torch.Generator().manual_seed(6))
    (out["comp_rgb_full"] * g.to(dev)).sum().backward()
    (full * g).sum().backward()
    lin = [m for m in model.texture.network.layers if isinstance(m, torch.nn.Linear)]
    for m, p in zip(lin, tex):
        assert rel_err(m.weight.grad, p["w"].grad) < 1e-3 and rel_err(m.bias.grad, p["b"].grad) < 1e-3
    gt = model.geometry.encoding.encoding.encoding.params.grad.cpu()
    assert F.cosine_similarity(gt[None], table.grad[None]).item() > 0.999


def test_neus_analytic_grad_type_is_honoured(dev):
    """grad_type 'analytic': sdf_grad_samples must equal the analytic gradient of the field (oracle/analytic.py),
    which differs measurably from the finite-difference gradient the fused path would have produced."""
    import rise_sdf_amd as R
    torch.manual_seed(1)
    cfg = model_config(hidden=32, n_levels=4, feat=13)
    cfg["geometry"]["grad_type"] = "analytic"
    cfg["geometry"]["finite_difference_eps"] = 1e-3
    model = R.make("neus", R.Config(dict(cfg))).to(dev)
    model.train()
    _prep(model, dev)
    assert model._general and not model._fused_ok()
    rays = camera_rays(12, 12, seed=7)
    u = torch.rand(rays.shape[0], generator=torch.Generator().manual_seed(8))
    out = model.forward_(rays.to(dev), stratified_u=u.to(dev))
    ri = out["ray_indices"].cpu()
    pts = rays[:, :3][ri] + rays[:, 3:][ri] * out["points"].cpu()[:, None]
    meta, table, mlp, _ = oracle_params(model)
    sdf_o, grad_o, _ = OA.volume_sdf_analytic(pts.double(), table.detach().double(), meta,
                                              [{k: v.detach().double() for k, v in p.items()} for p in mlp], radius=1.5)
    sdf_o, grad_o = sdf_o.detach(), grad_o.detach()
    assert rel_err(out["sdf_samples"], sdf_o) < 1e-5
    assert rel_err(out["sdf_grad_samples"], grad_o) < 1e-3
    # and it is NOT the finite-difference normal at eps = 1e-3 (what round 1 silently returned)
    _, fd_grad, _ = oracle.volume_sdf(pts, table.detach(), meta, [{k: v.detach() for k, v in p.items()} for p in mlp],
                                      radius=1.5, fd_eps=1e-3)
    assert rel_err(out["sdf_grad_samples"], grad_o) < 0.2 * rel_err(fd_grad, grad_o)
    out["opacity"].sum().backward()
    assert model.geometry.encoding.encoding.encoding.params.grad.abs().sum() > 0


def test_feature_query_full_position_jacobian(dev):
    """geometry(x, with_grad=False, input_grad=True): d feature / d x includes the xyz pass-through columns."""
    import rise_sdf_amd as R
    torch.manual_seed(2)
    geo = R.make("volume-sdf", model_config(hidden=32, n_levels=4, feat=13).geometry).to(dev)
    geo.train()
    with torch.no_grad():
        geo.encoding.encoding.encoding.params.mul_(300.0)
        l0 = geo.network.layers[0]
        l0.weight_v[:, 3:] = torch.randn_like(l0.weight_v[:, 3:]) * 0.3
    geo.update_step(0, 0)
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(500, 3, generator=g) * 2 - 1) * 1.2
    w = torch.randn(500, 13, generator=g)
    xd = x.to(dev).requires_grad_(True)
    f = geo(xd, with_grad=False, with_feature=True, input_grad=True)[1]
    (gx,) = torch.autograd.grad((f * w.to(dev)).sum(), xd)
    model = type("M", (), {"geometry": geo, "variance": type("V", (), {"variance": torch.tensor(0.3)})})
    meta, table, mlp, _ = oracle_params(model)
    x64 = x.double().requires_grad_(True)
    f_o = OA.field(x64, table.detach().double(), meta, [{k: v.detach().double() for k, v in p.items()} for p in mlp],
                   radius=1.5)
    (gx_o,) = torch.autograd.grad((f_o * w.double()).sum(), x64)
    assert rel_err(f, f_o) < 1e-5
    assert rel_err(gx, gx_o) < 1e-3
    # default call keeps skipping the pass-through columns (FD normals never need them): strictly a partial Jacobian
    xd2 = x.to(dev).requires_grad_(True)
    f2 = geo(xd2, with_grad=False, with_feature=True)[1]
    (gx2,) = torch.autograd.grad((f2 * w.to(dev)).sum(), xd2)
    assert rel_err(gx2, gx_o) > 10 * rel_err(gx, gx_o)
