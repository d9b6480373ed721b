"""B3-B7 on the device: the HIP path driven through the module-level entry points the reference binds to
(``nerfacc`` / ``nerfacc.volrend`` 0.5.3 names, the vendored ``_C`` names, ``tcnn.Encoding``, ``dr.texture``, the
renderutils plugin names), compared with the oracle.  The reference's files never travel here: what they would call
is exercised with the argument shapes their call sites use (tests/dropin_probe.py lists those sites)."""
import sys

import numpy as np
import pytest
import torch

import oracle
from oracle import envlight as E
from oracle import gridsample as OG
from oracle import texture as OT
from helpers import camera_rays, rel_err, sphere_binary

pytestmark = pytest.mark.gpu

ROI = torch.tensor([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5])


@pytest.fixture()
def mods(dev):
    import rise_sdf_amd.dropin as dropin
    saved = {k: sys.modules.get(k) for k in dropin.SLOTS}
    dropin.install(patch_renderutils=False)
    import nerfacc
    import nerfacc.volrend as volrend
    import nvdiffrast.torch as dr
    import tinycudann as tcnn
    C = sys.modules["lib.nerfacc.cuda._backend"]._C
    yield dict(nerfacc=nerfacc, volrend=volrend, dr=dr, tcnn=tcnn, C=C)
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


def test_vendored_C_marcher_and_grid(dev, mods):
    """lib/nerfacc/ray_marching.py:177-190 and grid.py:42-47 as the vendored Python would call them."""
    C = mods["C"]
    rays = camera_rays(24, 24, seed=3)
    o, d = rays[:, :3].contiguous(), rays[:, 3:].contiguous()
    binary = sphere_binary(32)
    tn, tf = oracle.ray_aabb_intersect(o, d, ROI)
    gtn, gtf = C.ray_aabb_intersect(o.to(dev), d.to(dev), ROI.to(dev))
    assert torch.equal(gtn.cpu(), tn) and torch.equal(gtf.cpu(), tf)
    pk, ri, ts, te = C.ray_marching(o.to(dev), d.to(dev), gtn, gtf, ROI.to(dev), binary.to(dev),
                                    C.ContractionType(0), 0.011, 0.0)
    rpk, rri, rts, rte = oracle.ray_marching_packed(o, d, tn, tf, ROI, binary, 0.011)
    assert pk.dtype == torch.int32 and ri.dtype == torch.int64 and ts.shape == (ri.numel(), 1)
    assert torch.equal(pk.cpu(), rpk) and torch.equal(ri.cpu(), rri)
    assert torch.equal(ts.cpu()[:, 0], rts.reshape(-1)) and torch.equal(te.cpu()[:, 0], rte.reshape(-1))
    assert torch.equal(C.unpack_info(pk, ri.numel()).cpu(), rri)
    # grid_query on float occupancies, points partly outside the box
    g = torch.Generator().manual_seed(0)
    x = (torch.rand(4096, 3, generator=g) * 2 - 1) * 1.7
    occs = torch.rand(32, 32, 32, generator=g)
    got = C.grid_query(x.to(dev), ROI.to(dev), occs.to(dev), C.ContractionType.AABB).cpu()
    inside, _ = oracle.query_occ(x, ROI, torch.ones(32, 32, 32, dtype=torch.bool))
    unit = (x - ROI[:3]) / (ROI[3:] - ROI[:3])
    ijk = (unit * 32).to(torch.int32).clamp(0, 31).long()
    want = torch.where(inside, occs[ijk[:, 0], ijk[:, 1], ijk[:, 2]], torch.zeros(()))
    assert torch.equal(got, want)
    xr = torch.rand(100, 3, generator=g)
    assert torch.allclose(C.contract(C.contract_inv(xr.to(dev), ROI.to(dev), C.ContractionType.AABB), ROI.to(dev),
                                     C.ContractionType.AABB).cpu(), xr, atol=1e-6)


def test_vendored_C_rendering_kernels(dev, mods):
    """lib/nerfacc/vol_rendering.py:303-307,430-434 docstring KATs through the _C names, forward and backward."""
    C = mods["C"]
    alphas = torch.tensor([[0.4], [0.8], [0.1], [0.8], [0.1], [0.0], [0.9]])
    ray_indices = torch.tensor([0, 0, 0, 1, 1, 2, 2])
    pk = oracle.pack_info(ray_indices, 3)
    w = C.weight_from_alpha_forward_naive(pk.to(dev), alphas.to(dev))
    t = C.transmittance_from_alpha_forward_naive(pk.to(dev), alphas.to(dev))
    assert w.shape == alphas.shape
    assert torch.allclose(w.cpu()[:, 0], torch.tensor([0.4, 0.48, 0.012, 0.8, 0.02, 0.0, 0.9]), atol=1e-6)
    assert torch.allclose(t.cpu()[:, 0], torch.tensor([1.0, 0.6, 0.12, 1.0, 0.2, 1.0, 1.0]), atol=1e-6)
    g = torch.Generator().manual_seed(1)
    a = (torch.rand(500, 1, generator=g) * 0.6).requires_grad_(True)
    ri = torch.sort(torch.randint(0, 40, (500,), generator=g))[0]
    pk = oracle.pack_info(ri, 40)
    gw = torch.randn(500, 1, generator=g)
    rw, rt = oracle.render_weight_from_alpha(a[:, 0], packed_info=pk)
    (ra,) = torch.autograd.grad(rw, a, gw[:, 0])
    wd = C.weight_from_alpha_forward_naive(pk.to(dev), a.detach().to(dev))
    ga = C.weight_from_alpha_backward_naive(wd, gw.to(dev), pk.to(dev), a.detach().to(dev))
    assert rel_err(wd[:, 0], rw) < 1e-6 and rel_err(ga, ra) < 1e-5
    rt2 = oracle.render_transmittance_from_alpha(a[:, 0], packed_info=pk)
    (rta,) = torch.autograd.grad(rt2, a, gw[:, 0])
    td = C.transmittance_from_alpha_forward_naive(pk.to(dev), a.detach().to(dev))
    gta = C.transmittance_from_alpha_backward_naive(pk.to(dev), a.detach().to(dev), td, gw.to(dev))
    assert rel_err(gta, rta) < 1e-5
    # density forms: alpha = 1 - exp(-sigma dt)
    ts = torch.rand(500, 1, generator=g)
    te = ts + 0.01 + 0.05 * torch.rand(500, 1, generator=g)
    sg = (torch.rand(500, 1, generator=g) * 20).requires_grad_(True)
    rws, _ = oracle.render_weight_from_alpha((1 - torch.exp(-sg * (te - ts)))[:, 0], packed_info=pk)
    (rsg,) = torch.autograd.grad(rws, sg, gw[:, 0])
    ws = C.weight_from_sigma_forward_naive(pk.to(dev), ts.to(dev), te.to(dev), sg.detach().to(dev))
    gs = C.weight_from_sigma_backward_naive(ws, gw.to(dev), pk.to(dev), ts.to(dev), te.to(dev), sg.detach().to(dev))
    assert rel_err(ws[:, 0], rws) < 1e-6 and rel_err(gs, rsg) < 1e-5


def test_nerfacc_053_names(dev, mods):
    """models/neus.py:164,195-197; models/volrend.py:227-233 (flat [S] tensors, keyword ray_indices / n_rays)."""
    nerfacc, volrend = mods["nerfacc"], mods["volrend"]
    rays = camera_rays(16, 16, seed=5)
    o, d = rays[:, :3].contiguous(), rays[:, 3:].contiguous()
    tn, tf = oracle.ray_aabb_intersect(o, d, ROI)
    t_mins, t_maxs, hits = nerfacc.ray_aabb_intersect(o.to(dev), d.to(dev), ROI.to(dev).reshape(1, 6))
    assert t_mins.shape == (256, 1) and hits.dtype == torch.bool
    hit_ref = tn < 1e10
    assert torch.equal(hits.cpu()[:, 0], hit_ref)
    assert torch.equal(t_maxs.cpu()[hit_ref, 0], tf[hit_ref]) and torch.equal(t_mins.cpu()[hit_ref, 0], tn[hit_ref])
    assert bool(torch.isinf(t_maxs.cpu()[~hit_ref]).all())
    g = torch.Generator().manual_seed(2)
    S, N = 700, 50
    ri = torch.sort(torch.randint(0, N, (S,), generator=g))[0]
    ts = torch.rand(S, generator=g)
    te = ts + 0.02
    sg = (torch.rand(S, generator=g) * 30).requires_grad_(True)
    sd = sg.detach().to(dev).requires_grad_(True)
    w, tr, al = volrend.render_weight_from_density(ts.to(dev), te.to(dev), sd, ray_indices=ri.to(dev), n_rays=N)
    ra = 1 - torch.exp(-sg * (te - ts))
    rw, rt = oracle.render_weight_from_alpha(ra, ray_indices=ri, n_rays=N)
    assert rel_err(w, rw) < 1e-6 and rel_err(tr, rt) < 1e-6 and rel_err(al, ra) < 1e-6
    vals = torch.randn(S, 3, generator=g)
    comp = volrend.accumulate_along_rays(w, values=vals.to(dev), ray_indices=ri.to(dev), n_rays=N)
    rcomp = oracle.accumulate_along_rays(rw, vals, ray_indices=ri, n_rays=N)
    go = torch.randn(N, 3, generator=g)
    (gsd,) = torch.autograd.grad(comp, sd, go.to(dev))
    (gsr,) = torch.autograd.grad(rcomp, sg, go)
    assert rel_err(comp, rcomp) < 1e-5 and rel_err(gsd, gsr) < 1e-4
    assert nerfacc.volrend is volrend and volrend.render_weight_from_alpha is nerfacc.render_weight_from_alpha


def test_tcnn_encoding_otypes(dev, mods):
    """models/network_utils.py:99 builds the direction encoding through tcnn.Encoding too."""
    tcnn = mods["tcnn"]
    sh = tcnn.Encoding(3, {"otype": "SphericalHarmonics", "degree": 5})
    g = torch.Generator().manual_seed(0)
    d = torch.nn.functional.normalize(torch.randn(2000, 3, generator=g), dim=-1)
    d01 = ((d + 1) / 2)
    x = d01.to(dev).requires_grad_(True)
    out = sh(x)
    xr = d01.double().requires_grad_(True)
    ref = OT.sh_encode(xr, 5)
    go = torch.randn(2000, 25, generator=g)
    (gx,) = torch.autograd.grad(out, x, go.to(dev))
    (gr,) = torch.autograd.grad(ref, xr, go.double())
    assert out.shape == (2000, 25) and rel_err(out, ref) < 1e-5 and rel_err(gx, gr) < 1e-4
    hg = tcnn.Encoding(3, {"otype": "HashGrid", "n_levels": 4, "n_features_per_level": 2, "log2_hashmap_size": 12,
                           "base_resolution": 8, "per_level_scale": 1.5}).to(dev)
    meta, n_params = oracle.grid_meta(4, 2, 12, 8, 1.5)
    assert hg.params.numel() == n_params and list(hg.state_dict()) == ["params"]
    xs = torch.rand(3000, 3, generator=g)
    assert torch.equal(hg(xs.to(dev)).cpu(), oracle.hashgrid_encode(xs, hg.params.detach().cpu(), meta))
    with pytest.raises(NotImplementedError):
        tcnn.Encoding(3, {"otype": "OneBlob"})


def test_dr_texture_2d_clamp(dev, mods):
    """models/texture.py:338-341: FG_LUT [1,256,256,2], uv [1,S,1,2], 'linear' + 'clamp', with the uv gradient and its
    second order (the path taken when normals come from an autograd graph that is differentiated again)."""
    dr = mods["dr"]
    lut = OT.synthetic_fg_lut(64)                                   # [1,64,64,2]
    g = torch.Generator().manual_seed(3)
    uv = torch.rand(1, 1500, 1, 2, generator=g) * 1.1 - 0.05       # a little outside [0,1]: clamp region
    t = lut.to(dev).requires_grad_(True)
    u = uv.to(dev).requires_grad_(True)
    out = dr.texture(t, u, filter_mode="linear", boundary_mode="clamp")
    assert out.shape == (1, 1500, 1, 2)
    t64 = lut.double().requires_grad_(True)
    u64 = uv.double().requires_grad_(True)
    ref = OG.grid_sample_2d(t64.permute(0, 3, 1, 2), u64 * 2 - 1, padding_mode="border",
                            align_corners=False).permute(0, 2, 3, 1)
    go = torch.randn(1, 1500, 1, 2, generator=g)
    gt, gu = torch.autograd.grad(out, (t, u), go.to(dev), create_graph=True)
    rt_, ru_ = torch.autograd.grad(ref, (t64, u64), go.double(), create_graph=True)
    assert rel_err(out, ref) < 1e-5 and rel_err(gt, rt_) < 1e-5 and rel_err(gu, ru_) < 1e-4
    v = torch.randn(1, 1500, 1, 2, generator=g)
    (g2,) = torch.autograd.grad((gu * v.to(dev)).sum(), t)
    (r2,) = torch.autograd.grad((ru_ * v.double()).sum(), t64)
    assert rel_err(g2, r2) < 1e-4
    with pytest.raises(NotImplementedError):
        dr.texture(t, u, filter_mode="linear", boundary_mode="wrap")


def test_dr_texture_cube(dev, mods):
    """lib/pbr/light.py:194-206: 'linear' on the diffuse map, 'linear-mipmap-linear' with an explicit mip list and
    mip_level_bias [1,S,1] on the specular stack; gradients to every level, the directions and the level."""
    dr = mods["dr"]
    g = torch.Generator().manual_seed(4)
    mips = [torch.rand(6, r, r, 3, generator=g) for r in (32, 16, 8)]
    S = 3000
    d = torch.randn(S, 3, generator=g)
    d[:6] = torch.tensor([[1.0, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]])
    d[6] = torch.tensor([1.0, 0.999, 0.2])
    lvl = torch.rand(S, generator=g) * 2.0
    lvl[:4] = torch.tensor([0.0, 1.0, 2.0, 1.5])
    md = [m.to(dev).requires_grad_(True) for m in mips]
    dd = d.to(dev).requires_grad_(True)
    ld = lvl.to(dev).requires_grad_(True)
    lin = dr.texture(md[2][None], dd.reshape(1, S, 1, 3).contiguous(), filter_mode="linear", boundary_mode="cube")
    assert lin.shape == (1, S, 1, 3)
    ref_lin = E.cube_sample_linear(mips[2].double(), d.double())
    assert rel_err(lin.reshape(S, 3), ref_lin) < 1e-5
    out = dr.texture(md[0][None], dd.reshape(1, S, 1, 3).contiguous(), mip=[m[None] for m in md[1:]],
                     mip_level_bias=ld.reshape(1, S, 1), filter_mode="linear-mipmap-linear", boundary_mode="cube")
    m64 = [m.double().requires_grad_(True) for m in mips]
    d64 = d.double().requires_grad_(True)
    l64 = lvl.double().requires_grad_(True)
    ref = E.cube_sample_mip(m64, d64, l64)
    go = torch.randn(S, 3, generator=g)
    grads = torch.autograd.grad(out.reshape(S, 3), md + [dd, ld], go.to(dev))
    rgrads = torch.autograd.grad(ref, m64 + [d64, l64], go.double())
    assert rel_err(out.reshape(S, 3), ref) < 1e-5
    for a, b in zip(grads[:3], rgrads[:3]):
        assert rel_err(a, b) < 1e-4
    # direction gradients: exclude samples whose bilinear footprint touches a texel border (kink)
    assert float((grads[3].cpu().double() - rgrads[3]).abs().median()) < 1e-5
    # d/d(level): one-sided at exactly integer levels (entries 0..2 were set to 0, 1, 2): compare the others
    assert rel_err(grads[4][4:], rgrads[4][4:]) < 1e-4


def test_renderutils_plugin_names(dev):
    """lib/renderutils/ops.py:391-458 running on the plugin-shaped object (same math as envlight.diffuse/specular)."""
    from rise_sdf_amd import renderutils as ru
    from rise_sdf_amd.envlight import ndf_cutoff
    g = torch.Generator().manual_seed(5)
    c = torch.rand(6, 16, 16, 3, generator=g)
    go = torch.randn(6, 16, 16, 3, generator=g)
    out = ru.plugin.diffuse_cubemap_fwd(c.to(dev))
    gc = ru.plugin.diffuse_cubemap_bwd(c.to(dev), go.to(dev))
    c64 = c.double().requires_grad_(True)
    ref = E.diffuse_cubemap(c64)
    (rc,) = torch.autograd.grad(ref, c64, go.double())
    assert rel_err(out, ref) < 1e-5 and rel_err(gc, rc) < 1e-5
    cosc = ndf_cutoff(1.0, 0.99)
    b = ru.plugin.specular_bounds(16, cosc)
    o4 = ru.plugin.specular_cubemap_fwd(c.to(dev), b, 1.0, cosc)
    assert o4.shape == (6, 16, 16, 4)
    spec = o4[..., :3] / o4[..., 3:]
    rs = E.specular_cubemap(c.double(), 1.0, 0.99)
    assert rel_err(spec, rs) < 1e-4
    assert rel_err(ru.specular_cubemap(c.to(dev), 1.0, 0.99), rs) < 1e-4
    g4 = torch.randn(6, 16, 16, 4, generator=g)
    gcs = ru.plugin.specular_cubemap_bwd(c.to(dev), b, g4.to(dev), 1.0, cosc)
    assert gcs.shape == (6, 16, 16, 3) and bool(torch.isfinite(gcs).all())
