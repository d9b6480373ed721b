"""GPU parity: every C-ABI op against the CPU oracle on identical seeded inputs.

Bars (BASELINE.json north_star): integer / index results bit-exact; fp32 values within 1e-4
relative; hash-table gradients (fp32 atomics, order-dependent) within 1e-3 rel of the fp64-accumulated
oracle.  Tolerances are written at each assert.
"""
import math

import numpy as np
import os

import pytest
import torch

import oracle
from helpers import camera_rays, rel_err, small_field, sphere_binary

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from rise_sdf_amd import ops as o
    return o


def test_library_is_loaded(dev):
    from rise_sdf_amd import _lib
    assert _lib.lib().rsdf_abi_version() == 3


# ---- M1 -------------------------------------------------------------------------------------------
def test_ray_aabb_bit_exact(dev, ops):
    rays = camera_rays(64, 64, seed=3)
    g = torch.Generator().manual_seed(0)
    extra_o = (torch.rand(500, 3, generator=g) * 2 - 1) * 3
    extra_d = torch.nn.functional.normalize(torch.randn(500, 3, generator=g), dim=-1)
    extra_d[:20, 0] = 0.0  # axis-parallel rays: division by zero must behave like the reference
    extra_d[20:40, 1] = 0.0
    o = torch.cat([rays[:, :3], extra_o])
    d = torch.cat([rays[:, 3:], extra_d])
    aabb = torch.tensor([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5])
    tn, tf = oracle.ray_aabb_intersect(o, d, aabb)
    gn, gf = ops.ray_aabb_intersect(o.to(dev), d.to(dev), aabb.to(dev))
    assert torch.equal(gn.cpu(), tn) and torch.equal(gf.cpu(), tf)


# ---- M3 / M4 --------------------------------------------------------------------------------------
@pytest.mark.parametrize("res,step", [(32, 0.00507421875), (128, 0.00507421875), (16, 0.05)])
def test_marcher_bit_exact(dev, ops, res, step):
    rays = camera_rays(48, 48, seed=res)
    o, d = rays[:, :3].contiguous(), rays[:, 3:].contiguous()
    roi = torch.tensor([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5])
    binary = sphere_binary(res)
    tn, tf = oracle.ray_aabb_intersect(o, d, roi)
    u = torch.rand(o.shape[0], generator=torch.Generator().manual_seed(2))
    tn = tn + u * step
    pk, ri, ts, te = oracle.ray_marching_packed(o, d, tn, tf, roi, binary, step)
    gpk, gri, gts, gte = ops.march(o.to(dev), d.to(dev), tn.to(dev), tf.to(dev), roi.to(dev),
                                   binary.to(dev), step)
    assert ri.numel() > 1000
    assert torch.equal(gpk.cpu(), pk), "packed_info differs"
    assert torch.equal(gri.cpu(), ri), "ray_indices differ"
    assert torch.equal(gts.cpu(), ts) and torch.equal(gte.cpu(), te), "sample intervals differ"


def test_marcher_dense_and_empty(dev, ops):
    rays = camera_rays(16, 16, seed=9)
    o, d = rays[:, :3].contiguous(), rays[:, 3:].contiguous()
    roi = torch.tensor([-1e10] * 3 + [1e10] * 3)
    ones = torch.ones(1, 1, 1, dtype=torch.bool)
    aabb = torch.tensor([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5])
    tn, tf = oracle.ray_aabb_intersect(o, d, aabb)
    pk, ri, ts, te = oracle.ray_marching_packed(o, d, tn, tf, roi, ones, 0.00507421875)
    gpk, gri, gts, gte = ops.march(o.to(dev), d.to(dev), tn.to(dev), tf.to(dev), roi.to(dev),
                                   ones.to(dev), 0.00507421875)
    assert torch.equal(gpk.cpu(), pk) and torch.equal(gts.cpu(), ts) and torch.equal(gte.cpu(), te)
    # all-empty grid -> zero samples, shapes still valid
    zeros = torch.zeros(8, 8, 8, dtype=torch.bool)
    gpk, gri, gts, gte = ops.march(o.to(dev), d.to(dev), tn.to(dev), tf.to(dev), aabb.to(dev),
                                   zeros.to(dev), 0.01)
    assert gri.numel() == 0 and int(gpk[:, 1].sum()) == 0


def test_query_occ_cells_bit_exact(dev, ops):
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(20000, 3, generator=g) * 2 - 1) * 1.6
    # points exactly on cell faces and on the box boundary
    grid = torch.linspace(-1.5, 1.5, 33)
    x[:33, 0] = grid
    x[33:66, 1] = grid
    roi = torch.tensor([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5])
    binary = sphere_binary(32)
    occ, cell = oracle.query_occ(x, roi, binary)
    gocc, gcell = ops.query_occ(x.to(dev), roi.to(dev), binary.to(dev), return_cell=True)
    assert torch.equal(gcell.cpu(), cell) and torch.equal(gocc.cpu(), occ)


# ---- M5 / M6 --------------------------------------------------------------------------------------
def _ragged(n_rays, max_len, seed, empty_every=5):
    g = torch.Generator().manual_seed(seed)
    counts = torch.randint(0, max_len, (n_rays,), generator=g)
    counts[::empty_every] = 0
    ri = torch.repeat_interleave(torch.arange(n_rays), counts)
    return counts, ri


def test_pack_info_of_the_marchers_own_ray_indices(dev, ops):
    """ops.pack_info returns the marcher's packed_info when it is handed the very ray_indices tensor the marcher returned
    (remembered by identity), and recomputes for any other tensor -- equal content either way; a copy, a slice, or an
    in-place change of the tensor all take the recomputation."""
    from helpers import camera_rays, sphere_binary
    rays = camera_rays(20, 20, seed=4)
    roi = torch.tensor([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5])
    binary = sphere_binary(32, 0.3, 0.9)
    ro, rd = rays[:, :3].contiguous().to(dev), rays[:, 3:].contiguous().to(dev)
    tmin, tmax = ops.ray_aabb_intersect(ro, rd, roi.to(dev))
    pk, ri, ts, te = ops.march(ro, rd, tmin, tmax, roi.to(dev), binary.to(dev), 3.0 / 256)
    assert ri.numel() > 0
    assert ops.pack_info(ri, ro.shape[0]) is pk
    again = ops.pack_info(ri.clone(), ro.shape[0])
    assert again is not pk and torch.equal(again, pk)
    assert ops.pack_info(ri, ro.shape[0] + 1) is not pk            # another ray count: recomputed
    assert torch.equal(pk.cpu(), oracle.pack_info(ri.cpu(), ro.shape[0]))
    ri.add_(0)                                                       # touched in place: no longer trusted
    assert ops.pack_info(ri, ro.shape[0]) is not pk


def test_pack_unpack_compact(dev, ops):
    counts, ri = _ragged(1000, 300, 1)
    pk = oracle.pack_info(ri, 1000)
    gpk = ops.pack_info(ri.to(dev), 1000)
    assert torch.equal(gpk.cpu(), pk)
    assert torch.equal(ops.unpack_info(gpk, ri.numel()).cpu(), ri)
    g = torch.Generator().manual_seed(2)
    keep = torch.rand(ri.numel(), generator=g) > 0.4
    ts = torch.rand(ri.numel(), generator=g)
    te = ts + 0.1
    r2, s2, e2 = ops.compact_samples(keep.to(dev), ri.to(dev), ts.to(dev), te.to(dev))
    assert torch.equal(r2.cpu(), ri[keep]) and torch.equal(s2.cpu(), ts[keep]) and torch.equal(e2.cpu(), te[keep])
    # nothing kept / empty input
    r3, _, _ = ops.compact_samples(torch.zeros_like(keep).to(dev), ri.to(dev), ts.to(dev), te.to(dev))
    assert r3.numel() == 0
    assert ops.pack_info(torch.zeros(0, dtype=torch.int64, device=dev), 7).cpu().tolist() == [[0, 0]] * 7


# ---- C1 / C2 --------------------------------------------------------------------------------------
def test_compositing_kat(dev, ops):
    """Docstring known answers: lib/nerfacc/vol_rendering.py:303-307, 430-434, 493-500."""
    a = torch.tensor([0.4, 0.8, 0.1, 0.8, 0.1, 0.0, 0.9], device=dev)
    ri = torch.tensor([0, 0, 0, 1, 1, 2, 2], device=dev)
    w, t = ops.render_weight_from_alpha(a, ray_indices=ri, n_rays=3)
    assert torch.allclose(t.cpu(), torch.tensor([1.0, 0.6, 0.12, 1.0, 0.2, 1.0, 1.0]), atol=1e-6)
    assert torch.allclose(w.cpu(), torch.tensor([0.4, 0.48, 0.012, 0.8, 0.02, 0.0, 0.9]), atol=1e-6)
    vis = ops.render_visibility(a, ray_indices=ri, n_rays=3, early_stop_eps=0.3, alpha_thre=0.2)
    assert vis.cpu().tolist() == [True, True, False, True, False, False, True]


@pytest.mark.parametrize("max_len", [40, 700])
def test_weights_fwd_bwd(dev, ops, max_len):
    counts, ri = _ragged(300, max_len, 3)
    g = torch.Generator().manual_seed(4)
    a = torch.rand(ri.numel(), generator=g) * 0.3
    a[::97] = 1.0  # fully opaque samples: the 1e-10 clamp path of the backward
    a[5::131] = 0.0
    gw = torch.randn(ri.numel(), generator=g)
    a_o = a.clone().requires_grad_(True)
    w_o, t_o = oracle.render_weight_from_alpha(a_o, ray_indices=ri, n_rays=300)
    (w_o * gw).sum().backward()
    a_g = a.to(dev).requires_grad_(True)
    w_g, t_g = ops.render_weight_from_alpha(a_g, ray_indices=ri.to(dev), n_rays=300)
    (w_g * gw.to(dev)).sum().backward()
    # fp32, scan order differs from the serial reference loop: 1e-5 relative
    assert rel_err(w_g, w_o) < 1e-5 and rel_err(t_g, t_o) < 1e-5
    # At alpha == 1 exactly the reference computes (gw*T - sum_{k>=j} gw_k w_k) / 1e-10 where the
    # numerator is 0 in exact arithmetic: its value there is fp32 rounding residue x 1e10 (the serial
    # oracle shows +-600), i.e. noise, so those entries are only required to be finite.
    mask = a < 1.0
    assert torch.allclose(a_g.grad.cpu()[mask], a_o.grad[mask], rtol=1e-4, atol=1e-5)
    assert bool(torch.isfinite(a_g.grad).all())
    # transmittance backward
    a_o2 = a.clone().requires_grad_(True)
    (oracle.render_transmittance_from_alpha(a_o2, ray_indices=ri, n_rays=300) * gw).sum().backward()
    a_g2 = a.to(dev).requires_grad_(True)
    (ops.render_transmittance_from_alpha(a_g2, ray_indices=ri.to(dev), n_rays=300) * gw.to(dev)).sum().backward()
    assert torch.allclose(a_g2.grad.cpu()[mask], a_o2.grad[mask], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("D", [1, 3, 7, 24])
def test_accumulate_fwd_bwd(dev, ops, D):
    counts, ri = _ragged(200, 150, 6)
    g = torch.Generator().manual_seed(7)
    w = torch.rand(ri.numel(), generator=g)
    v = torch.randn(ri.numel(), D, generator=g)
    go = torch.randn(200, D, generator=g)
    w_o, v_o = w.clone().requires_grad_(True), v.clone().requires_grad_(True)
    out_o = oracle.accumulate_along_rays(w_o, v_o, ray_indices=ri, n_rays=200)
    (out_o * go).sum().backward()
    w_g, v_g = w.to(dev).requires_grad_(True), v.to(dev).requires_grad_(True)
    out_g = ops.accumulate_along_rays(w_g, v_g, ray_indices=ri.to(dev), n_rays=200)
    (out_g * go.to(dev)).sum().backward()
    assert torch.allclose(out_g.cpu(), out_o, rtol=1e-5, atol=1e-5)
    assert torch.allclose(w_g.grad.cpu(), w_o.grad, rtol=1e-5, atol=1e-5)
    assert torch.allclose(v_g.grad.cpu(), v_o.grad, rtol=1e-6, atol=1e-6)
    if D == 1:  # values=None path
        o2 = ops.accumulate_along_rays(w.to(dev), None, ray_indices=ri.to(dev), n_rays=200)
        assert torch.allclose(o2.cpu(), oracle.accumulate_along_rays(w, None, ray_indices=ri, n_rays=200),
                              rtol=1e-5, atol=1e-5)


def test_opacity_and_depth_in_one_pass(dev, ops):
    """models/volrend.py:878-885: ops.accumulate_opacity_depth == the two accumulate_along_rays calls bit for bit (values
    and the weight gradient), == the oracle within fp32 rounding; rays without samples included; the optional midpoints
    output equals (t_starts + t_ends) / 2."""
    counts, ri = _ragged(300, 150, 9)
    g = torch.Generator().manual_seed(11)
    w = torch.rand(ri.numel(), generator=g)
    ts = torch.rand(ri.numel(), generator=g) * 3.0
    te = ts + torch.rand(ri.numel(), generator=g) * 0.01
    g_op, g_d = torch.randn(300, 1, generator=g), torch.randn(300, 1, generator=g)
    rid, tsd, ted = ri.to(dev), ts.to(dev), te.to(dev)
    w1 = w.to(dev).requires_grad_(True)
    op1 = ops.accumulate_along_rays(w1, None, ray_indices=rid, n_rays=300)
    d1 = ops.accumulate_along_rays(w1, (tsd + ted)[..., None] / 2.0, ray_indices=rid, n_rays=300)
    ((op1 * g_op.to(dev)).sum() + (d1 * g_d.to(dev)).sum()).backward()
    w2 = w.to(dev).requires_grad_(True)
    op2, d2, mid = ops.accumulate_opacity_depth(w2, tsd, ted, ray_indices=rid, n_rays=300, want_midpoints=True)
    ((op2 * g_op.to(dev)).sum() + (d2 * g_d.to(dev)).sum()).backward()
    assert torch.equal(op2, op1) and torch.equal(d2, d1)
    assert torch.equal(w2.grad, w1.grad)
    assert torch.equal(mid, (tsd + ted) / 2.0) and not mid.requires_grad
    w_o = w.clone().requires_grad_(True)
    op_o = oracle.accumulate_along_rays(w_o, None, ray_indices=ri, n_rays=300)
    d_o = oracle.accumulate_along_rays(w_o, (ts + te)[..., None] / 2.0, ray_indices=ri, n_rays=300)
    ((op_o * g_op).sum() + (d_o * g_d).sum()).backward()
    assert torch.allclose(op2.cpu(), op_o.detach(), rtol=1e-5, atol=1e-5)
    assert torch.allclose(d2.cpu(), d_o.detach(), rtol=1e-5, atol=1e-5)
    assert torch.allclose(w2.grad.cpu(), w_o.grad, rtol=1e-5, atol=1e-5)
    w3 = w.to(dev).requires_grad_(True)                       # only one of the two outputs is used downstream
    op3, d3 = ops.accumulate_opacity_depth(w3, tsd, ted, ray_indices=rid, n_rays=300)
    (d3 * g_d.to(dev)).sum().backward()
    w4 = w.to(dev).requires_grad_(True)
    (ops.accumulate_along_rays(w4, (tsd + ted)[..., None] / 2.0, ray_indices=rid, n_rays=300) * g_d.to(dev)).sum().backward()
    assert torch.equal(w3.grad, w4.grad)


def test_opacity_depth_and_normal_map_in_one_pass(dev, ops):
    """models/volrend.py:875-885: ops.accumulate_opacity_depth_normal == accumulate_opacity_depth + accumulate_along_rays on
    the [S,3] normals, bit for bit (values, weight gradient, normal gradient); rays without samples included; partial
    gradient sets; sample arrays with an unowned tail (capacity mode) get zero gradients there."""
    counts, ri = _ragged(300, 150, 19)
    g = torch.Generator().manual_seed(21)
    S = ri.numel()
    w, nm = torch.rand(S, generator=g), torch.randn(S, 3, generator=g)
    ts = torch.rand(S, generator=g) * 3.0
    te = ts + torch.rand(S, generator=g) * 0.01
    g_op, g_d, g_n = (torch.randn(300, 1, generator=g).to(dev), torch.randn(300, 1, generator=g).to(dev),
                      torch.randn(300, 3, generator=g).to(dev))
    rid, tsd, ted = ri.to(dev), ts.to(dev), te.to(dev)
    for use in ((0, 0, 1), (1, 0, 0), (0, 1, 1), (1, 1, 1)):
        w1, n1 = w.to(dev).requires_grad_(True), nm.to(dev).requires_grad_(True)
        op1, d1, mid1 = ops.accumulate_opacity_depth(w1, tsd, ted, ray_indices=rid, n_rays=300, want_midpoints=True)
        m1 = ops.accumulate_along_rays(w1, n1, ray_indices=rid, n_rays=300)
        (use[0] * (op1 * g_op).sum() + use[1] * (d1 * g_d).sum() + use[2] * (m1 * g_n).sum()).backward()
        w2, n2 = w.to(dev).requires_grad_(True), nm.to(dev).requires_grad_(True)
        op2, d2, m2, mid2 = ops.accumulate_opacity_depth_normal(w2, tsd, ted, n2, ray_indices=rid, n_rays=300,
                                                                want_midpoints=True)
        terms = [t for u, t in zip(use, ((op2 * g_op).sum(), (d2 * g_d).sum(), (m2 * g_n).sum())) if u]
        sum(terms).backward()
        assert torch.equal(op2, op1) and torch.equal(d2, d1) and torch.equal(m2, m1) and torch.equal(mid2, mid1)
        assert torch.equal(w2.grad, w1.grad), use
        if use[2]:
            assert torch.equal(n2.grad, n1.grad), use
        else:
            assert n2.grad is None or not bool(n2.grad.any())
    assert bool((m2[counts.to(dev) == 0] == 0).all())
    # an unowned tail: 40 more samples than the rays' packed_info covers (march_capped's dummy tail)
    pk = ops.pack_info(rid, 300)
    pad = lambda t: torch.cat([t, torch.full((40,) + t.shape[1:], 0.5, device=dev)])
    w3, n3 = pad(w.to(dev)).requires_grad_(True), pad(nm.to(dev)).requires_grad_(True)
    op3, d3, m3 = ops.accumulate_opacity_depth_normal(w3, pad(tsd), pad(ted), n3, packed_info=pk)
    ((op3 * g_op).sum() + (d3 * g_d).sum() + (m3 * g_n).sum()).backward()
    assert torch.equal(op3, op1) and torch.equal(m3, m1)
    assert bool((w3.grad[S:] == 0).all()) and bool((n3.grad[S:] == 0).all()) and torch.equal(w3.grad[:S], w2.grad)


# ---- H1 -------------------------------------------------------------------------------------------
GRIDS = [dict(n_levels=4, n_features=2, log2_hashmap_size=14, base_resolution=16, per_level_scale=1.5),
         dict(n_levels=16, n_features=2, log2_hashmap_size=19, base_resolution=32,
              per_level_scale=1.447269237440378)]


@pytest.mark.parametrize("gi", [0, 1])
def test_hashgrid_forward_bit_exact(dev, ops, gi):
    from rise_sdf_amd import _lib
    cfg = GRIDS[gi]
    meta_o, n_params = oracle.grid_meta(**cfg)
    meta_g, n_params_g = _lib.make_grid_meta(**cfg)
    assert n_params == n_params_g
    g = torch.Generator().manual_seed(11)
    table = (torch.rand(n_params, generator=g) * 2 - 1) * 1e-4
    x = torch.rand(5000, 3, generator=g)
    x[:4] = torch.tensor([[0, 0, 0], [1, 1, 1], [0.5, 0.5, 0.5], [1, 0, 1]])  # box corners / faces
    ref = oracle.hashgrid_encode(x, table, meta_o)
    out = ops.hashgrid_encode(x.to(dev), table.to(dev), meta_g)
    assert torch.equal(out.cpu(), ref), f"max abs diff {float((out.cpu() - ref).abs().max())}"
    # progressive mask + include_xyz (H2)
    ref2 = oracle.composite_encoding(x, table, meta_o, n_active_levels=3)
    out2 = ops.hashgrid_encode(x.to(dev), table.to(dev), meta_g, n_active_levels=3, include_xyz=True)
    assert torch.equal(out2.cpu(), ref2)


def _entry_points_of(fn):
    """(fn(), the set of C-ABI entry points it launched)."""
    from rise_sdf_amd import _lib
    timer = _lib.KernelTimer()
    _lib.set_timer(timer)
    try:
        out = fn()
    finally:
        _lib.set_timer(None)
    return out, {r[0] for r in timer.records}


@pytest.mark.parametrize("gi", [0, 1])
@pytest.mark.parametrize("n", [1, 127, 5000, (1 << 15) + 77])
def test_hashgrid_forward_staged_bit_exact(dev, ops, gi, n, monkeypatch):
    """rsdf_hashgrid_fwd_staged (level-major planes + rows through LDS; what large batches take) writes the rows of
    rsdf_hashgrid_fwd bit for bit: plain, with the progressive mask + include_xyz, ragged tile ends; and the oracle's
    values where the oracle is cheap."""
    from rise_sdf_amd import _lib
    cfg = GRIDS[gi]
    meta_o, n_params = oracle.grid_meta(**cfg)
    meta_g, _ = _lib.make_grid_meta(**cfg)
    g = torch.Generator().manual_seed(13)
    table = ((torch.rand(n_params, generator=g) * 2 - 1) * 1e-4).to(dev)
    x = torch.rand(n, 3, generator=g)
    x[:1] = torch.tensor([[1.0, 0.0, 1.0]])
    xd = x.to(dev)
    for kw in (dict(), dict(n_active_levels=3, include_xyz=True), dict(n_active_levels=0, include_xyz=True)):
        monkeypatch.setenv("RSDF_GATHER", "rows")
        rows = ops.hashgrid_encode(xd, table, meta_g, **kw)
        monkeypatch.setenv("RSDF_GATHER", "staged")
        staged, calls = _entry_points_of(lambda: ops.hashgrid_encode(xd, table, meta_g, **kw))
        assert calls == {"rsdf_hashgrid_fwd_staged"}
        assert torch.equal(staged, rows), kw
    monkeypatch.delenv("RSDF_GATHER")
    out, calls = _entry_points_of(lambda: ops.hashgrid_encode(xd, table, meta_g))
    assert calls == ({"rsdf_hashgrid_fwd_staged"} if n >= 1 << 15 else {"rsdf_hashgrid_fwd"})   # the default's threshold
    if n <= 5000:
        assert torch.equal(out.cpu(), oracle.hashgrid_encode(x, table.cpu(), meta_o))


def test_hashgrid_forward_staged_several_passes(dev, ops, monkeypatch):
    """More points than one pass of the staged gather holds (2^22): the passes' rows join up bit for bit."""
    from rise_sdf_amd import _lib
    meta_g, n_params = _lib.make_grid_meta(**GRIDS[1])
    g = torch.Generator().manual_seed(14)
    table = ((torch.rand(n_params, generator=g) * 2 - 1) * 1e-4).to(dev)
    x = torch.rand((1 << 22) + 333, 3, generator=g).to(dev)
    monkeypatch.setenv("RSDF_GATHER", "rows")
    rows = ops.hashgrid_encode(x, table, meta_g, include_xyz=True)
    monkeypatch.setenv("RSDF_GATHER", "staged")
    staged = ops.hashgrid_encode(x, table, meta_g, include_xyz=True)
    assert torch.equal(staged, rows)


@pytest.mark.parametrize("gi", [0, 1])
def test_hashgrid_backward(dev, ops, gi):
    from rise_sdf_amd import _lib
    cfg = GRIDS[gi]
    meta_o, n_params = oracle.grid_meta(**cfg)
    meta_g, _ = _lib.make_grid_meta(**cfg)
    g = torch.Generator().manual_seed(12)
    table = (torch.rand(n_params, generator=g) * 2 - 1) * 1e-4
    # ray-like coherent points (runs of equal cells) plus random ones
    t = torch.linspace(0, 1, 3000)[:, None]
    x = torch.cat([0.1 + 0.8 * t * torch.tensor([[0.7, 0.5, 0.3]]), torch.rand(2000, 3, generator=g)])
    gout = torch.randn(x.shape[0], meta_o.n_levels * 2, generator=g)
    t_o = table.clone().requires_grad_(True)
    (oracle.hashgrid_encode(x, t_o, meta_o) * gout).sum().backward()
    t_g = table.to(dev).requires_grad_(True)
    (ops.hashgrid_encode(x.to(dev), t_g, meta_g) * gout.to(dev)).sum().backward()
    # fp32 atomics vs fp64-accumulated oracle: 1e-3 relative to the largest row, 1e-7 absolute
    scale = float(t_o.grad.abs().max())
    assert float((t_g.grad.cpu() - t_o.grad).abs().max()) < 1e-5 * scale + 1e-7
    assert int((t_g.grad.cpu() != 0).sum()) == int((t_o.grad != 0).sum())


# ---- H3 -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("K,N,act", [(35, 64, "softplus100"), (64, 64, "softplus100"), (64, 48, "none"),
                                     (35, 128, "softplus100"), (128, 128, "relu"), (128, 48, "none"),
                                     (11, 32, "softplus100"), (32, 13, "none"), (84, 128, "relu"),
                                     (128, 3, "sigmoid")])
def test_linear_fwd_bwd(dev, ops, K, N, act):
    g = torch.Generator().manual_seed(K * 1000 + N)
    n = 1000  # not a multiple of the 128-row tile
    x = torch.randn(n, K, generator=g) * 0.5
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g) * 0.1
    gy = torch.randn(n, N, generator=g)

    def ref_act(z):
        return {"softplus100": lambda t: torch.nn.functional.softplus(t, beta=100),
                "relu": torch.relu, "none": lambda t: t, "sigmoid": torch.sigmoid}[act](z)

    xo, wo, bo = [t.clone().double().requires_grad_(True) for t in (x, w, b)]
    yo = ref_act(torch.nn.functional.linear(xo, wo, bo))
    (yo * gy.double()).sum().backward()
    xg, wg, bg = [t.to(dev).requires_grad_(True) for t in (x, w, b)]
    yg = ops.linear(xg, wg, bg, act=act)
    (yg * gy.to(dev)).sum().backward()
    # fp32 MFMA (exact fp32 fma chain) vs an fp64 reference: 1e-5 relative to the tensor's scale
    assert rel_err(yg, yo) < 1e-5
    assert rel_err(xg.grad, xo.grad) < 1e-5
    assert rel_err(wg.grad, wo.grad) < 2e-5  # fp32 atomics over 1000 rows
    assert rel_err(bg.grad, bo.grad) < 2e-5


def test_linear_column_window(dev, ops):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(300, 35, generator=g)
    w = torch.randn(64, 35, generator=g) / 6
    b = torch.zeros(64)
    gy = torch.randn(300, 64, generator=g)
    xo = x.clone().requires_grad_(True)
    (torch.nn.functional.softplus(torch.nn.functional.linear(xo, w, b), beta=100) * gy).sum().backward()
    xg = x.to(dev).requires_grad_(True)
    (ops.linear(xg, w.to(dev), b.to(dev), act="softplus100", dx_cols=(3, 32)) * gy.to(dev)).sum().backward()
    assert torch.allclose(xg.grad.cpu()[:, 3:], xo.grad[:, 3:], rtol=1e-4, atol=1e-5)
    assert float(xg.grad[:, :3].abs().max()) == 0.0


def test_weight_norm(dev, ops):
    g = torch.Generator().manual_seed(2)
    v = torch.randn(64, 35, generator=g)
    gg = torch.rand(64, 1, generator=g) + 0.5
    dw = torch.randn(64, 35, generator=g)
    vo, go = v.clone().requires_grad_(True), gg.clone().requires_grad_(True)
    (oracle.weight_norm_effective(go, vo) * dw).sum().backward()
    vg, g2 = v.to(dev).requires_grad_(True), gg.to(dev).requires_grad_(True)
    w = ops.weight_norm(g2, vg)
    (w * dw.to(dev)).sum().backward()
    assert torch.allclose(w.cpu(), oracle.weight_norm_effective(gg, v), rtol=1e-6, atol=1e-7)
    assert torch.allclose(vg.grad.cpu(), vo.grad, rtol=1e-4, atol=1e-6)
    assert torch.allclose(g2.grad.cpu(), go.grad, rtol=1e-4, atol=1e-6)


# ---- golden fixtures through the HIP path ------------------------------------------------------------
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_vanilla_mlp_golden(dev, ops, golden_dir, tag):
    """Reference VanillaMLP outputs/grads (tests/golden/make_golden.py) vs the HIP layers."""
    import os
    z = np.load(os.path.join(golden_dir, f"vanilla_mlp_{tag}.npz"))
    x = torch.tensor(z["x"], device=dev, requires_grad=True)
    P = {}
    for i in (0, 2, 4):
        P[i] = [torch.tensor(z[f"layers_{i}_{n}"], device=dev, requires_grad=True)
                for n in ("weight_g", "weight_v", "bias")]
    h = x
    for i in (0, 2, 4):
        gpar, v, b = P[i]
        h = ops.linear(h, ops.weight_norm(gpar, v), b, act="none" if i == 4 else "softplus100")
    assert rel_err(h, torch.tensor(z["y"])) < 1e-5
    (h * torch.tensor(z["gy"], device=dev)).sum().backward()
    assert rel_err(x.grad, torch.tensor(z["gx"])) < 1e-4
    for i in (0, 2, 4):
        for t, n in zip(P[i], ("weight_g", "weight_v", "bias")):
            assert rel_err(t.grad, torch.tensor(z[f"grad_layers_{i}_{n}"])) < 1e-4, (i, n)


def test_get_alpha_golden(dev, ops, golden_dir):
    import os
    z = np.load(os.path.join(golden_dir, "get_alpha.npz"))
    sdf, nrm, dirs, dists = [torch.tensor(z[k], device=dev) for k in ("sdf", "normal", "dirs", "dists")]
    for vi, v in enumerate(z["variances"]):
        for ci, c in enumerate(z["cos_anneal"]):
            a = ops.neus_alpha(sdf, nrm, dirs, dists, torch.tensor(float(v), device=dev), float(c))
            ref = torch.tensor(z[f"alpha_v{vi}_c{ci}"])
            # 1e-4 relative (north_star) on values that are not clipped to 0
            assert torch.allclose(a.cpu(), ref, rtol=1e-4, atol=1e-6), (vi, ci)


# ---- A1 / H4 backward --------------------------------------------------------------------------------
def test_neus_alpha_backward(dev, ops):
    g = torch.Generator().manual_seed(8)
    S = 4000
    sdf = torch.randn(S, generator=g) * 0.05
    nrm = torch.nn.functional.normalize(torch.randn(S, 3, generator=g), dim=-1)
    dirs = torch.nn.functional.normalize(torch.randn(S, 3, generator=g), dim=-1)
    dists = torch.full((S, 1), 0.005) * (1 + torch.rand(S, 1, generator=g))
    ga = torch.randn(S, generator=g)
    for car in (1.0, 0.3):
        var = torch.tensor(0.3)
        so, no, vo = sdf.clone().requires_grad_(True), nrm.clone().requires_grad_(True), var.clone().requires_grad_(True)
        a_o = oracle.get_alpha(so, no, dirs, dists, oracle.inv_s_from_variance(vo), car)
        (a_o * ga).sum().backward()
        sg, ng, vg = sdf.to(dev).requires_grad_(True), nrm.to(dev).requires_grad_(True), var.to(dev).requires_grad_(True)
        a_g = ops.neus_alpha(sg, ng, dirs.to(dev), dists.to(dev), vg, car)
        (a_g * ga.to(dev)).sum().backward()
        assert torch.allclose(a_g.cpu(), a_o, rtol=1e-4, atol=1e-6)
        assert rel_err(sg.grad, so.grad) < 1e-4
        assert rel_err(ng.grad, no.grad) < 1e-4
        assert abs(float(vg.grad) - float(vo.grad)) < 1e-3 * abs(float(vo.grad)) + 1e-4


def _same_zero_pattern(got, ref, scale):
    """The binned table scatter's queue records hold an entry's two feature gradients as a block-float pair with 20
    significant bits (hashgrid_fd7.hip PairRec): a feature whose contributions are all below 2^-21 of their partner's
    quantises to exactly zero (with Gaussian gradients: ~6e-7 of the single-record entries).  So the sets of touched
    entries may differ in a few entries per million, and only where the reference value is itself negligible."""
    mism = (got != 0) != (ref != 0)
    n = int(mism.sum())
    assert n <= max(4, int(2e-5 * int((ref != 0).sum()))), n
    if n:
        assert float(ref[mism].abs().max()) <= 1e-5 * scale and float(got[mism].abs().max()) <= 1e-5 * scale


# ---- H1b, finite-difference stencil variant (bin + LDS reduce, no per-corner atomics) ----------------
@pytest.mark.parametrize("gi,eps_unit", [(1, 1.0 / 8192), (1, 1.0 / 1291), (0, 1.0 / 54), (1, 3.0 / 8192)])
def test_hashgrid_backward_fd7(dev, ops, gi, eps_unit):
    """Same gradient as the generic scatter on [S,7,3] stencil points: compared with the oracle's
    fp64-accumulated backward.  eps = one finest cell (the progressive schedule), a mid level's cell,
    and 3 finest cells (taps more than one cell away: the slow-path branch)."""
    from rise_sdf_amd import _lib
    cfg = GRIDS[gi]
    meta_o, n_params = oracle.grid_meta(**cfg)
    meta_g, _ = _lib.make_grid_meta(**cfg)
    g = torch.Generator().manual_seed(21)
    table = (torch.rand(n_params, generator=g) * 2 - 1) * 1e-4
    S = 3000
    t = torch.linspace(0, 1, S // 2)[:, None]
    centre = torch.cat([0.05 + 0.9 * t * torch.tensor([[0.9, 0.6, 0.35]]),
                        torch.rand(S - S // 2, 3, generator=g)])
    centre[:3] = torch.tensor([[0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [0.5, 0.0, 1.0]])  # clamped taps
    offs = torch.tensor([[0, 0, 0], [1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]],
                        dtype=torch.float32) * eps_unit
    x7 = (centre[:, None, :] + offs[None]).clamp(0.0, 1.0).reshape(-1, 3).contiguous()
    LF = meta_o.n_levels * 2
    gout = torch.randn(x7.shape[0], 3 + LF, generator=g)
    t_o = table.clone().requires_grad_(True)
    (oracle.hashgrid_encode(x7, t_o, meta_o) * gout[:, 3:]).sum().backward()
    t_g = table.to(dev).requires_grad_(True)
    enc = ops.hashgrid_encode(x7.to(dev), t_g, meta_g, include_xyz=True, fd7_eps_unit=eps_unit)
    (enc * gout.to(dev)).sum().backward()
    scale = float(t_o.grad.abs().max())
    err = float((t_g.grad.cpu() - t_o.grad).abs().max())
    assert err < 1e-5 * scale + 1e-7, (err, scale)
    _same_zero_pattern(t_g.grad.cpu(), t_o.grad, scale)


@pytest.mark.parametrize("S,n_levels,n_active,write_xyz", [(1, 16, 16, 1), (33, 16, 9, 1), (1000, 4, 4, 0), (4097, 16, 16, 1)])
def test_stencil_layout_kernels_are_the_permuted_copies(dev, S, n_levels, n_active, write_xyz):
    """rsdf_stencil_points_tap_major / _planes_to_rows / _rows_to_planes against the torch expressions they replace in the
    reference-shaped stencil entry (ops._HashGrid): bit-identical, ragged sample counts, masked levels, with and without
    the xyz columns, rows wider than the written window."""
    import ctypes
    from rise_sdf_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(S)
    x7 = torch.rand(S, 7, 3, generator=g).to(dev)
    x7t = torch.empty(7, S, 3, device=dev)
    assert L.rsdf_stencil_points_tap_major(_lib.ptr(x7), S, _lib.ptr(x7t), _lib.stream_ptr()) == 0
    assert torch.equal(x7t, x7.permute(1, 0, 2).contiguous())
    planes = torch.randn(n_levels, 7, S, 2, generator=g).to(dev)
    col = 3 if write_xyz else 2                     # (a column offset that is not the xyz width either)
    ld = col + 2 * n_levels + 1
    out = torch.full((7 * S, ld), -7.0, device=dev)
    assert L.rsdf_stencil_planes_to_rows(_lib.ptr(planes), _lib.ptr(x7), S, n_levels, n_active, _lib.ptr(out), ld, col,
                                         write_xyz, 2.0, -1.0, _lib.stream_ptr()) == 0
    want = torch.full((7 * S, ld), -7.0, device=dev)
    pm = planes.clone()
    pm[n_active:] = 0.0
    want[:, col:col + 2 * n_levels] = pm.permute(2, 1, 0, 3).reshape(7 * S, 2 * n_levels)
    if write_xyz:
        want[:, :3] = x7.view(-1, 3) * 2.0 + -1.0
    assert torch.equal(out, want)
    grad = torch.randn(7 * S, ld, generator=g).to(dev)
    dpl = torch.empty(n_levels, 7, S, 2, device=dev)
    assert L.rsdf_stencil_rows_to_planes(_lib.ptr(grad), ld, col, S, n_levels, _lib.ptr(dpl), _lib.stream_ptr()) == 0
    assert torch.equal(dpl, grad[:, col:col + 2 * n_levels].reshape(S, 7, n_levels, 2).permute(2, 1, 0, 3).contiguous())


def test_hashgrid_forward_fd7_bit_exact(dev, ops):
    """Stencil-merged forward (one gather of the centre cell + 4 corners per displaced tap) must equal
    the generic per-point encoding bit for bit, including taps clamped at the box and eps > one cell."""
    import ctypes
    from rise_sdf_amd import _lib
    cfg = GRIDS[1]
    meta_o, n_params = oracle.grid_meta(**cfg)
    meta_g, _ = _lib.make_grid_meta(**cfg)
    g = torch.Generator().manual_seed(31)
    table = (torch.rand(n_params, generator=g) * 2 - 1) * 1e-4
    S = 2500
    for eps_unit in (1.0 / 8192, 1.0 / 1291, 3.0 / 8192):
        centre = torch.rand(S, 3, generator=g)
        centre[:3] = torch.tensor([[0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [0.5, 0.0, 1.0]])
        offs = torch.tensor([[0, 0, 0], [1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]],
                            dtype=torch.float32) * eps_unit
        x7t = (centre[None, :, :] + offs[:, None, :]).clamp(0.0, 1.0).contiguous()  # [7,S,3]
        ref = oracle.hashgrid_encode(x7t.reshape(-1, 3), table, meta_o).view(7, S, 16, 2)
        planes = torch.empty(16, 7, S, 2, device=dev)
        xg, tg = x7t.to(dev), table.to(dev)
        rc = _lib.lib().rsdf_hashgrid_fwd_fd7(_lib.ptr(xg), _lib.ptr(tg), ctypes.byref(meta_g), S, 16,
                                              _lib.ptr(planes), _lib.stream_ptr())
        assert rc == 0
        assert torch.equal(planes.cpu().permute(1, 2, 0, 3), ref), eps_unit


# ---- H1 / H1b stencil kernels with the taps DERIVED in-kernel from world-space centres ---------------------------
def _stencil_points(dev, ops, S, eps, radius=1.5, seed=41):
    """World-space centres incl. box faces, points an ulp outside the box (the marcher's rounding) and far outside,
    pushed through rsdf_fd_points as degenerate rays (o = p, d = 0) -> (x7t [7,S,3], pts [S,3]) on the device."""
    g = torch.Generator().manual_seed(seed)
    p = (torch.rand(S, 3, generator=g) * 2 - 1) * radius
    up = float(np.nextafter(np.float32(radius), np.float32(2 * radius)))
    p[:8] = torch.tensor([[radius, radius, radius], [-radius, -radius, -radius], [up, 0.0, -up], [-up, up, 0.3],
                          [radius - eps, -radius + eps, 0.0], [radius - eps / 2, 0.0, -radius + eps / 2],
                          [1.7, 1.9, 0.2], [0.0, 0.0, 0.0]])
    o = p.to(dev)
    d = torch.zeros_like(o)
    ri = torch.arange(S, device=dev)
    t = torch.zeros(S, device=dev)
    return ops.fd_points(o, d, ri, t, t, radius, eps, want_positions=True, tap_major=True)


@pytest.mark.parametrize("eps_cells", [1.0, 6.35, 3.0])
def test_hashgrid_fd7_pts_forward_bit_exact(dev, ops, eps_cells):
    """rsdf_hashgrid_fwd_fd7_pts (stencil derived from the centre) == rsdf_hashgrid_fwd_fd7 on
    the x7t that rsdf_fd_points wrote for the same samples, bit for bit -- and therefore == the generic encoder."""
    import ctypes
    from rise_sdf_amd import _lib
    cfg = GRIDS[1]
    meta_o, n_params = oracle.grid_meta(**cfg)
    meta_g, _ = _lib.make_grid_meta(**cfg)
    radius, S = 1.5, 5000
    eps = 2 * radius / 8192 * eps_cells        # one finest cell, a mid level's cell, 3 finest cells (slow path)
    tg = ((torch.rand(n_params, generator=torch.Generator().manual_seed(5)) * 2 - 1) * 1e-4).to(dev)
    x7t, pts = _stencil_points(dev, ops, S, eps, radius)
    assert torch.equal(pts, pts) and x7t.shape == (7, S, 3)
    a = torch.empty(16, 7, S, 2, device=dev)
    b = torch.full_like(a, float("nan"))
    assert _lib.lib().rsdf_hashgrid_fwd_fd7(_lib.ptr(x7t), _lib.ptr(tg), ctypes.byref(meta_g), S, 16, _lib.ptr(a),
                                            _lib.stream_ptr()) == 0
    assert _lib.lib().rsdf_hashgrid_fwd_fd7_pts(_lib.ptr(pts), radius, eps, _lib.ptr(tg), ctypes.byref(meta_g), S, 16,
                                                _lib.ptr(b), _lib.stream_ptr()) == 0
    assert torch.equal(a, b)
    # and both equal the oracle wherever the whole stencil lies in the unit cube (the oracle's encoder is only defined there;
    # the out-of-box centres above exist to pin the two kernels' clamping against each other)
    xc = x7t.cpu()
    inside = ((xc >= 0) & (xc <= 1)).all(-1).all(0)
    assert int(inside.sum()) > S - 16
    ref = oracle.hashgrid_encode(xc[:, inside].reshape(-1, 3), tg.cpu(), meta_o).view(7, -1, 16, 2)
    assert torch.equal(b.cpu().permute(1, 2, 0, 3)[:, inside], ref)


@pytest.mark.parametrize("eps_cells", [1.0, 6.35, 3.0])
def test_hashgrid_fd7_pts_backward(dev, ops, eps_cells):
    """The derived-stencil backward against the oracle's fp64-accumulated scatter and against the x7t form."""
    import ctypes
    from rise_sdf_amd import _lib
    cfg = GRIDS[1]
    meta_o, n_params = oracle.grid_meta(**cfg)
    meta_g, _ = _lib.make_grid_meta(**cfg)
    radius, S = 1.5, 6000
    eps = 2 * radius / 8192 * eps_cells
    eps_unit = eps / (2 * radius)
    x7t, pts = _stencil_points(dev, ops, S, eps, radius, seed=43)
    dpl = torch.randn(16, 7, S, 2, generator=torch.Generator().manual_seed(7)).to(dev)
    nbytes = int(_lib.lib().rsdf_hashgrid_bwd_fd7_scratch_bytes(ctypes.byref(meta_g), S, 16, eps_unit))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    d_x = torch.zeros(n_params, device=dev)
    d_p = torch.zeros(n_params, device=dev)
    assert _lib.lib().rsdf_hashgrid_bwd_fd7(_lib.ptr(x7t), _lib.ptr(dpl), ctypes.byref(meta_g), S, 16, eps_unit,
                                            _lib.ptr(d_x), _lib.ptr(scratch), nbytes, _lib.stream_ptr()) == 0
    assert _lib.lib().rsdf_hashgrid_bwd_fd7_pts(_lib.ptr(pts), radius, eps, _lib.ptr(dpl), ctypes.byref(meta_g), S, 16,
                                                eps_unit, _lib.ptr(d_p), _lib.ptr(scratch), nbytes,
                                                _lib.stream_ptr()) == 0
    scale = float(d_x.abs().max())
    assert float((d_p - d_x).abs().max()) < 2e-6 * scale      # same records, different summation order only
    assert int((d_p != 0).sum()) == int((d_x != 0).sum())
    # against the oracle's fp64-accumulated scatter, on the samples whose whole stencil lies in the unit cube
    xc = x7t.cpu()
    inside = ((xc >= 0) & (xc <= 1)).all(-1).all(0)
    idx = inside.nonzero().view(-1).to(dev)
    Si = int(idx.numel())
    d_i = torch.zeros(n_params, device=dev)
    pts_i, dpl_i = pts[idx].contiguous(), dpl[:, :, idx].contiguous()
    assert _lib.lib().rsdf_hashgrid_bwd_fd7_pts(_lib.ptr(pts_i), radius, eps, _lib.ptr(dpl_i), ctypes.byref(meta_g), Si,
                                                16, eps_unit, _lib.ptr(d_i), _lib.ptr(scratch), nbytes,
                                                _lib.stream_ptr()) == 0
    t_o = torch.zeros(n_params, requires_grad=True)
    gout = dpl_i.cpu().permute(1, 2, 0, 3).reshape(7, Si, 32)        # [tap, sample, level * 2 + feature]
    (oracle.hashgrid_encode(xc[:, inside].reshape(-1, 3), t_o, meta_o).view(7, Si, 32) * gout).sum().backward()
    scale = float(t_o.grad.abs().max())
    assert float((d_i.cpu() - t_o.grad).abs().max()) < 1e-5 * scale + 1e-7
    _same_zero_pattern(d_i.cpu(), t_o.grad, scale)


# ---- the binned table scatter for plain points (generic backward, and the input gradient's backward) ---------------------
@pytest.mark.parametrize("mode", [0, 1])
def test_hashgrid_scatter_binned_matches_atomics(dev, ops, mode):
    """rsdf_hashgrid_scatter_binned against the per-corner atomic kernels it replaces above 16384 points: same table
    gradient (fp64-accumulated bins vs fp32 atomics: 1e-5 of the largest row), same set of touched entries; ray-like
    points (consecutive samples share coarse cells: the run merge) plus random ones."""
    import ctypes
    from rise_sdf_amd import _lib
    cfg = GRIDS[1]
    meta_g, n_params = _lib.make_grid_meta(**cfg)
    g = torch.Generator().manual_seed(51 + mode)
    n = 40000
    t = torch.linspace(0, 1, n // 2)[:, None]
    x = torch.cat([0.05 + 0.9 * t * torch.tensor([[0.9, 0.6, 0.35]]), torch.rand(n - n // 2, 3, generator=g)]).to(dev)
    table = ((torch.rand(n_params, generator=g) * 2 - 1) * 1e-2).to(dev)
    dy = torch.randn(n, 3 + 32, generator=g).to(dev)
    gdx = torch.randn(n, 3, generator=g).to(dev)
    L = _lib.lib()
    ref = torch.zeros(n_params, device=dev)
    if mode == 0:
        assert L.rsdf_hashgrid_bwd(_lib.ptr(x), _lib.ptr(dy), ctypes.byref(meta_g), n, 16, 35, 3, _lib.ptr(ref),
                                   _lib.stream_ptr()) == 0
    else:
        assert L.rsdf_hashgrid_dx_bwd(_lib.ptr(x), _lib.ptr(table), ctypes.byref(meta_g), n, 16, _lib.ptr(dy), 35, 3,
                                      _lib.ptr(gdx), None, 35, 3, _lib.ptr(ref), None, _lib.stream_ptr()) == 0
    nbytes = int(L.rsdf_hashgrid_scatter_binned_scratch_bytes(ctypes.byref(meta_g), n, 16))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    got = torch.zeros(n_params, device=dev)
    assert L.rsdf_hashgrid_scatter_binned(mode, _lib.ptr(x), _lib.ptr(dy), 35, 3, _lib.ptr(gdx) if mode else None,
                                          ctypes.byref(meta_g), n, 16, _lib.ptr(got), _lib.ptr(scratch), nbytes,
                                          _lib.stream_ptr()) == 0
    scale = float(ref.abs().max())
    assert scale > 0 and float((got - ref).abs().max()) < 2e-5 * scale
    _same_zero_pattern(got, ref, scale)          # (+ an entry whose contributions cancel to exactly 0 in one order)


def test_binned_scatter_record_format_bound(dev, ops):
    """The queue records (hashgrid_fd7.hip PairRec) carry an entry's two feature gradients as a block-float pair: 20
    significant bits for the larger one, the same absolute step for the smaller.  Per table entry the binned result must
    therefore lie within 2^-20 (half a step of a value just above a power of two; + three fp32 roundings of 2^-24: the
    product, the fp64 -> fp32 flush, the table add) x sum over its records of max(|v0|, |v1|) of the fp64 sum -- checked
    against the oracle with a second backward that yields exactly that bound; a feature 2^-30 of its partner quantises to zero
    without disturbing the partner; a non-finite gradient still poisons its rows."""
    import ctypes
    from rise_sdf_amd import _lib
    cfg = GRIDS[1]
    meta_o, n_params = oracle.grid_meta(**cfg)
    meta_g, _ = _lib.make_grid_meta(**cfg)
    g = torch.Generator().manual_seed(77)
    n = 2000
    x = torch.rand(n, 3, generator=g)
    dy = torch.randn(n, 32, generator=g) * torch.logspace(-6, 3, n)[:, None]       # nine decades of row magnitudes
    dy[::7, 1::2] *= 2.0 ** -30                                                        # feature 1 far below feature 0
    L = _lib.lib()

    def binned(dyt):
        nbytes = int(L.rsdf_hashgrid_scatter_binned_scratch_bytes(ctypes.byref(meta_g), n, 16))
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        got = torch.zeros(n_params, device=dev)
        dyd = dyt.to(dev).contiguous()
        assert L.rsdf_hashgrid_scatter_binned(0, _lib.ptr(x.to(dev)), _lib.ptr(dyd), 32, 0, None, ctypes.byref(meta_g), n, 16,
                                              _lib.ptr(got), _lib.ptr(scratch), nbytes, _lib.stream_ptr()) == 0
        return got.cpu()

    got = binned(dy)
    t = torch.zeros(n_params, dtype=torch.float64, requires_grad=True)
    (oracle.hashgrid_encode(x, t, meta_o).double() * dy.double()).sum().backward()
    ref = t.grad.clone()
    big = torch.maximum(dy[:, 0::2].abs(), dy[:, 1::2].abs()).repeat_interleave(2, dim=1)    # max(|d0|, |d1|) per level
    t.grad = None
    (oracle.hashgrid_encode(x, t, meta_o).double() * big.double()).sum().backward()
    bound = t.grad * 2.0 ** -19.6 + 1e-37
    assert bool(((got.double() - ref).abs() <= bound).all()), float(((got.double() - ref).abs() / bound).max())
    assert float((got.double() - ref).abs().max()) > 0            # (it IS a quantised format: not bit-equal to fp64)
    dy_bad = dy.clone()
    dy_bad[5, 4] = float("inf")
    bad = binned(dy_bad)
    assert int(torch.isnan(bad).sum()) >= 8 and int(torch.isnan(bad).sum()) <= 16     # the 8 corners of level 2, both features


def test_record_format_is_a_run_time_choice(dev, ops):
    """the hash backward's queue records -- block-float pairs (20 significant bits, the default) or fp32
    values (what the reference's fp32 atomics accumulate, models/network_utils.py:47-59) -- are both compiled in and chosen
    with rsdf_set_record_format / RSDF_REC=fp32, no rebuild.  The plain-point scatter and the stencil backward under both:
    the fp32 form within fp32 rounding of the fp64 oracle sum (2^-22 of the per-entry sum of |contributions|), the
    block-float form within its own bound and measurably coarser; scratch sizes follow the format (20 vs 16 bytes)."""
    import ctypes
    from rise_sdf_amd import _lib
    cfg = GRIDS[1]
    meta_o, n_params = oracle.grid_meta(**cfg)
    meta_g, _ = _lib.make_grid_meta(**cfg)
    g = torch.Generator().manual_seed(78)
    n = 3000
    x = torch.rand(n, 3, generator=g)
    dy = torch.randn(n, 32, generator=g) * torch.logspace(-4, 2, n)[:, None]
    L = _lib.lib()
    assert L.rsdf_get_record_format() == (1 if os.environ.get("RSDF_REC") == "fp32" else 0)
    t = torch.zeros(n_params, dtype=torch.float64, requires_grad=True)
    (oracle.hashgrid_encode(x, t, meta_o).double() * dy.double()).sum().backward()
    ref = t.grad.clone()
    t.grad = None
    (oracle.hashgrid_encode(x, t, meta_o).double() * dy.abs().double()).sum().backward()
    mass = t.grad.clone()                                # per entry: sum of |contributions|
    t.grad = None
    big = torch.maximum(dy[:, 0::2].abs(), dy[:, 1::2].abs()).repeat_interleave(2, dim=1)    # max(|d0|, |d1|) per level:
    (oracle.hashgrid_encode(x, t, meta_o).double() * big.double()).sum().backward()        # the block-float pair's step
    mass_pair = t.grad.clone()
    errs, sizes = {}, {}
    start = L.rsdf_get_record_format()
    try:
        for fmt in (0, 1):
            assert L.rsdf_set_record_format(fmt) == 0 and L.rsdf_get_record_format() == fmt
            nbytes = int(L.rsdf_hashgrid_scatter_binned_scratch_bytes(ctypes.byref(meta_g), n, 16))
            sizes[fmt] = nbytes
            scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            got = torch.zeros(n_params, device=dev)
            dyd = dy.to(dev).contiguous()
            assert L.rsdf_hashgrid_scatter_binned(0, _lib.ptr(x.to(dev)), _lib.ptr(dyd), 32, 0, None, ctypes.byref(meta_g), n, 16,
                                                  _lib.ptr(got), _lib.ptr(scratch), nbytes, _lib.stream_ptr()) == 0
            torch.cuda.synchronize()
            e = (got.cpu().double() - ref).abs()
            errs[fmt] = float((e / (mass_pair + 1e-300)).max())
            bound = (mass * 2.0 ** -22 if fmt else mass_pair * 2.0 ** -19.6) + 1e-37
            assert bool((e <= bound).all()), (fmt, float((e / bound).max()))
            # the stencil backward (the fused field's table gradient) under the same setting: against the atomics form
            from test_gpu_x2 import _field_inputs
            meta7, table7, ws7, x7t, pts, radius, eps = _field_inputs(dev, ops, 2000, 64, 13, seed=5)
            dpl = torch.randn(16, 7, 2000, 2, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
            nb7 = int(L.rsdf_hashgrid_bwd_fd7_scratch_bytes(ctypes.byref(meta7), 2000, 16, float(eps / (2 * radius))))
            sc7 = torch.empty(nb7, dtype=torch.uint8, device=dev)
            dt = torch.zeros_like(table7.detach())
            assert L.rsdf_hashgrid_bwd_fd7_pts(_lib.ptr(pts), float(radius), float(eps), _lib.ptr(dpl), ctypes.byref(meta7), 2000, 16,
                                               float(eps / (2 * radius)), _lib.ptr(dt), _lib.ptr(sc7), nb7, _lib.stream_ptr()) == 0
            torch.cuda.synchronize()
            errs[("fd7", fmt)] = dt.clone()
    finally:
        L.rsdf_set_record_format(start)
    assert sizes[1] > sizes[0] and abs((sizes[1] - sizes[0]) / sizes[0] - 0.25) < 0.05, sizes      # 20 against 16 bytes per element
    print(f"record formats vs the fp64 sum, worst entry / its |mass|: block-float {errs[0]:.2e}, fp32 values {errs[1]:.2e}")
    assert errs[1] < 2.0 ** -22 and errs[1] < errs[0]
    a, b = errs[("fd7", 0)], errs[("fd7", 1)]
    assert float((a - b).abs().max()) <= 2.0 ** -18 * float(b.abs().max()) and float((a - b).abs().max()) > 0


def test_large_hashmap_backward_falls_back_to_atomics(dev, ops):
    """a tcnn.Encoding with log2_hashmap_size = 21 (hashed levels beyond the bins' 2^19 entries) must still
    back-propagate above the binned scatter's point threshold -- through the per-corner atomic kernels -- and agree with
    the oracle's fp64-accumulated scatter."""
    from rise_sdf_amd import _lib
    cfg = dict(n_levels=8, n_features=2, log2_hashmap_size=21, base_resolution=32, per_level_scale=1.6)
    meta_g, n_params = _lib.make_grid_meta(**cfg)
    meta_o, n_o = oracle.grid_meta(**cfg)
    assert n_params == n_o
    n = 20000
    assert n >= ops.BINNED_SCATTER_MIN_POINTS and not ops._use_binned(meta_g, n, 8)
    g = torch.Generator().manual_seed(71)
    x = torch.rand(n, 3, generator=g)
    table = ((torch.rand(n_params, generator=g) * 2 - 1) * 1e-2)
    gout = torch.randn(n, 16, generator=g)
    tg = table.to(dev).requires_grad_(True)
    xg = x.to(dev).requires_grad_(True)
    y = ops.hashgrid_encode(xg, tg, meta_g)
    (y * gout.to(dev)).sum().backward()
    to = table.clone().requires_grad_(True)
    (oracle.hashgrid_encode(x, to, meta_o) * gout).sum().backward()
    scale = float(to.grad.abs().max())
    assert float((tg.grad.cpu() - to.grad).abs().max()) < 2e-5 * scale
    assert xg.grad is not None and bool(torch.isfinite(xg.grad).all())


def test_curvature_path_uses_binned_scatter_and_matches(dev, ops, monkeypatch):
    """ops.hashgrid_encode / ops.hashgrid_dx on 20000 points: the autograd backward with the binned scatter equals the one
    with RSDF_SCATTER=atomics."""
    from rise_sdf_amd import _lib
    cfg = GRIDS[1]
    meta_g, n_params = _lib.make_grid_meta(**cfg)
    g = torch.Generator().manual_seed(61)
    n = 20000
    x = torch.rand(n, 3, generator=g).to(dev)
    t0 = ((torch.rand(n_params, generator=g) * 2 - 1) * 1e-2).to(dev)
    go = torch.randn(n, 35, generator=g).to(dev)
    u = torch.randn(n, 35, generator=g).to(dev)
    gd = torch.randn(n, 3, generator=g).to(dev)
    res = []
    for mode in ("bins", "atomics"):
        monkeypatch.setenv("RSDF_SCATTER", mode)
        tb = t0.clone().requires_grad_(True)
        enc = ops.hashgrid_encode(x, tb, meta_g, include_xyz=True)
        dx = ops.hashgrid_dx(x, tb, u, meta_g, None, 3)
        ((enc * go).sum() + (dx * gd).sum()).backward()
        res.append(tb.grad.clone())
    scale = float(res[1].abs().max())
    assert float((res[0] - res[1]).abs().max()) < 2e-5 * scale


@pytest.mark.parametrize("res,fill,step,cone", [(32, 0.3, 0.00507421875, 0.0), (16, 0.5, 0.02, 0.004), (1, 1.0, 0.00507421875, 0.0)])
def test_staged_marcher_equals_the_two_passes(dev, ops, res, fill, step, cone):
    """ops.march / march_capped with a ``t_range_hint`` march every ray ONCE (the count pass parks the samples, the write pass
    copies them): same packed_info, ray_indices and (t0, t1) bit for bit as the reference-shaped two passes -- for a hint that
    covers every ray (the ROI diagonal), one that covers only the short rays (the others are marched again) and a useless one;
    through the capacity path with a truncating capacity too."""
    rays = camera_rays(40, 40, seed=res + 3)
    o, d = rays[:, :3].contiguous().to(dev), rays[:, 3:].contiguous().to(dev)
    roi = torch.tensor([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5]).to(dev)
    g = torch.Generator().manual_seed(res)
    binary = (torch.rand(res, res, res, generator=g) < fill).to(dev)
    tn, tf = ops.ray_aabb_intersect(o, d, roi)
    tn = tn + torch.rand(o.shape[0], generator=g).to(dev) * step
    ref = ops.march(o, d, tn, tf, roi, binary, step, cone)
    assert ref[1].numel() > 2000
    for hint in (5.2, 1.0, 1e-3):
        got = ops.march(o, d, tn, tf, roi, binary, step, cone, t_range_hint=hint)
        for a, b, name in zip(got, ref, ("packed_info", "ray_indices", "t_starts", "t_ends")):
            assert torch.equal(a, b), (name, hint)
    total = ref[1].numel()
    for cap in (total + 100, total // 2):
        base = ops.march_capped(o, d, tn, tf, roi, binary, step, cap, cone)
        for hint in (5.2, 1.0):
            got = ops.march_capped(o, d, tn, tf, roi, binary, step, cap, cone, t_range_hint=hint)
            for a, b, name in zip(got, base, ("packed_info", "ray_indices", "t_starts", "t_ends", "total")):
                assert torch.equal(a, b), (name, hint, cap)


@pytest.mark.parametrize("res,fill,step,cone", [(32, 0.3, 0.00507421875, 0.0), (64, 0.08, 0.0152631578947, 0.0),
                                                (16, 0.5, 0.02, 0.004), (128, 0.02, 0.00507421875, 0.0)])
def test_marcher_bit_exact_fragmented_grid(dev, ops, res, fill, step, cone):
    """The wave-per-ray marcher on grids where occupied and empty cells alternate at random (every few steps a skip, runs
    of every length, speculation depth changing all the time), with near / far clamps and a cone angle: packed_info,
    ray_indices and every (t0, t1) equal the oracle's thread-per-ray loop bit for bit."""
    rays = camera_rays(40, 40, seed=res + 1)
    o, d = rays[:, :3].contiguous(), rays[:, 3:].contiguous()
    roi = torch.tensor([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5])
    g = torch.Generator().manual_seed(res)
    binary = torch.rand(res, res, res, generator=g) < fill
    tn, tf = oracle.ray_aabb_intersect(o, d, roi)
    tn = torch.clamp(tn + torch.rand(o.shape[0], generator=g) * step, min=2.7)       # a near plane inside the box
    tf = torch.clamp(tf, max=5.2)
    pk, ri, ts, te = oracle.ray_marching_packed(o, d, tn, tf, roi, binary, step, cone)
    gpk, gri, gts, gte = ops.march(o.to(dev), d.to(dev), tn.to(dev), tf.to(dev), roi.to(dev), binary.to(dev), step, cone)
    assert ri.numel() > 2000
    assert torch.equal(gpk.cpu(), pk), "packed_info differs"
    assert torch.equal(gri.cpu(), ri), "ray_indices differ"
    assert torch.equal(gts.cpu(), ts) and torch.equal(gte.cpu(), te), "sample intervals differ"


def test_fd7_points_radius_range_is_checked(dev, ops):
    """The derived stencil takes the contraction's division by 2 r as multiply-adds that round like the division only while
    nothing over- or underflows (csrc/hashgrid_common.h unit_div): radii outside (2^-101, 2^99) are RSDF_EINVAL, not garbage."""
    import ctypes
    from rise_sdf_amd import _lib
    L, P = _lib.lib(), _lib.ptr
    meta, n_params = _lib.make_grid_meta(n_levels=4, n_features=2, log2_hashmap_size=12, base_resolution=8, per_level_scale=1.5)
    table = torch.zeros(int(n_params), device=dev)
    pts = torch.zeros(8, 3, device=dev)
    planes = torch.empty(4, 7, 8, 2, device=dev)
    for radius in (0.0, 1e-38, 1e38, float("inf")):
        rc = L.rsdf_hashgrid_fwd_fd7_pts(P(pts), radius, 1e-3, P(table), ctypes.byref(meta), 8, 4, P(planes), _lib.stream_ptr())
        assert rc != 0 and b"radius" in L.rsdf_last_error()
    assert L.rsdf_hashgrid_fwd_fd7_pts(P(pts), 1.5, 1e-3, P(table), ctypes.byref(meta), 8, 4, P(planes), _lib.stream_ptr()) == 0


def test_unit_div_rounds_like_the_division_exhaustively(dev):
    """tools/unit_div_check.hip on this GPU: unit_div (five multiply-adds) against the IEEE division for all 2^32 x and 18
    divisors (2 r of the yamls' radius 1.5 first); exit code 0 = no in-contract difference."""
    import shutil
    import subprocess
    import tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "unit_div_check")
        subprocess.check_call([hipcc, "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-w", "-I", os.path.join(root, "rise_sdf_amd", "csrc"),
                               "-I", os.path.join(root, "include"), "-o", exe, os.path.join(root, "tools", "unit_div_check.hip")])
        out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:]
    assert out.stdout.count("in-contract mismatches 0") == 18
