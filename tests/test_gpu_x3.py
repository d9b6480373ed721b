"""The pre-split input image of the fused SDF-MLP kernels (round 4: rsdf_hashgrid_fwd_fd7_x3, rsdf_sdfmlp_fd7_fwd_x3 / _bwd_x3).

The image must be EXACTLY the three-way bf16 split of what the fp32 stencil gather writes (bit-exact vs the oracle elsewhere),
and the MLP kernels that consume it must give what the kernels on the fp32 planes give: the split is deterministic, so the
matrix operands are the same bf16 numbers either way."""
import ctypes

import numpy as np
import pytest
import torch

import oracle  # noqa: F401  (conftest path)
from test_gpu_ops import GRIDS, _stencil_points

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from rise_sdf_amd import ops as o
    return o


def _bf16_bits_to_f32(u16):
    return (u16.to(torch.int32) << 16).view(torch.float32)


def _split3(x):
    """The kernels' split (split_bf16.h split3_pair): round to nearest even at every step."""
    h = x.to(torch.bfloat16).to(torch.float32)
    r = x - h
    m = r.to(torch.bfloat16).to(torch.float32)
    l = (r - m).to(torch.bfloat16).to(torch.float32)
    return h, m, l


@pytest.mark.parametrize("S,n_active,form", [(5000, 16, "pts"), (4097, 16, "x7t"), (777, 5, "pts"), (31, 16, "pts")])
def test_x3_image_is_the_exact_split_of_the_planes(dev, ops, S, n_active, form):
    from rise_sdf_amd import _lib
    L = _lib.lib()
    cfg = GRIDS[1]
    meta_g, n_params = _lib.make_grid_meta(**cfg)
    radius = 1.5
    eps = 2 * radius / 8192
    xyz_scale, xyz_offset = 2.0, -1.0
    tg = ((torch.rand(n_params, generator=torch.Generator().manual_seed(5)) * 2 - 1) * 1e-2).to(dev)
    x7t, pts = _stencil_points(dev, ops, S, eps, radius)
    planes = torch.zeros(16, 7, S, 2, device=dev)
    assert L.rsdf_hashgrid_fwd_fd7_pts(_lib.ptr(pts), radius, eps, _lib.ptr(tg), ctypes.byref(meta_g), S, n_active,
                                       _lib.ptr(planes), _lib.stream_ptr()) == 0
    Sp = int(L.rsdf_x3_rows(S))
    assert Sp % 32 == 0 and 0 <= Sp - S < 32 and int(L.rsdf_x3_bytes(S)) == 7 * 3 * 36 * Sp * 2
    x3 = torch.full((Sp // 32, 7, 3, 36, 32), 0x7FC0, dtype=torch.int16, device=dev)  # NaN patterns: every slot must be written
    assert L.rsdf_hashgrid_fwd_fd7_x3(_lib.ptr(x7t) if form == "x7t" else None, _lib.ptr(pts) if form == "pts" else None,
                                      radius, eps, _lib.ptr(tg), ctypes.byref(meta_g), S, n_active, xyz_scale, xyz_offset,
                                      _lib.ptr(x3), _lib.stream_ptr()) == 0
    torch.cuda.synchronize()
    # [tile][tap][part][column][32 rows], the row halves of columns with bit 3 set swapped -> [7, 3, 36, Sp]
    swapped = ((torch.arange(36, device=dev) >> 3) & 1).bool()
    x3 = torch.where(swapped[None, None, None, :, None], torch.cat([x3[..., 16:], x3[..., :16]], dim=-1), x3)
    parts = _bf16_bits_to_f32(x3).permute(1, 2, 3, 0, 4).reshape(7, 3, 36, Sp)
    assert bool((parts[..., S:] == 0).all()), "rows past n_samples must be zeros"
    assert bool((parts[:, 0, 35, :S] == 1).all()) and bool((parts[:, 1:, 35, :S] == 0).all()), "bias column"
    # hash-feature columns: column 2 l + f of tap t = planes[l, t, :, f]
    want = planes.permute(1, 0, 3, 2).reshape(7, 32, S)                             # [tap, 2 l + f, S]
    h, m, l = _split3(want)
    got = parts[:, :, :32, :S]
    assert torch.equal(got[:, 0], h) and torch.equal(got[:, 1], m) and torch.equal(got[:, 2], l)
    assert torch.equal((got[:, 0] + got[:, 1]) + got[:, 2], want), "h + m + l must give the fp32 value back exactly"
    if n_active < 16:
        assert bool((got[:, :, 2 * n_active:] == 0).all())
    # xyz columns: the tap's unit-cube coordinates (what rsdf_fd_points wrote) * scale + offset
    xyz = (x7t * xyz_scale + xyz_offset).permute(0, 2, 1)                           # [7, 3, S]
    hx, mx, lx = _split3(xyz)
    gx = parts[:, :, 32:35, :S]
    assert torch.equal(gx[:, 0], hx) and torch.equal(gx[:, 1], mx) and torch.equal(gx[:, 2], lx)


def _field_inputs(dev, ops, S, H, N2, seed=3, table_scale=3e-2):
    from rise_sdf_amd import _lib
    cfg = GRIDS[1]
    meta_g, n_params = _lib.make_grid_meta(**cfg)
    g = torch.Generator().manual_seed(seed)
    radius = 1.5
    eps = 2 * radius / 8192
    table = ((torch.rand(n_params, generator=g) * 2 - 1) * table_scale).to(dev).requires_grad_(True)
    K0 = 35
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev).requires_grad_(True)
    ws = [(mk(H, K0, sc=0.4), mk(H, sc=0.1)), (mk(H, H, sc=0.25), mk(H, sc=0.1)), (mk(N2, H, sc=0.3), mk(N2, sc=0.1))]
    x7t, pts = _stencil_points(dev, ops, S, eps, radius, seed=seed + 10)
    return meta_g, table, ws, x7t, pts, radius, eps


@pytest.mark.parametrize("S,n_active,want_feature", [(4133, 16, True), (2048, 16, False), (1000, 7, True)])
def test_x3_field_matches_the_planes_path(dev, ops, S, n_active, want_feature, monkeypatch):
    """rise_sdf_amd.fused.sdf_field_fd7 at H = 64 through the x3 image against the same node on the fp32 planes: the quad
    backward consumes the same bf16 operands in the same order (gradients equal to fp32 rounding of different atomics
    order); the forward's columns are summed in another order (1e-6)."""
    from rise_sdf_amd import fused
    H, N2 = 64, 13
    meta, table, ws, x7t, pts, radius, eps = _field_inputs(dev, ops, S, H, N2)
    eps_unit = eps / (2 * radius)
    outs = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("RSDF_X3", flag)
        assert fused.use_x3(35, H, N2, "fp32") == (flag == "1")
        for t in [table] + [p for wb in ws for p in wb]:
            t.grad = None
        sdf7t, feat = fused.sdf_field_fd7(x7t, table, ws, meta, n_active, 2.0, -1.0, eps_unit, want_feature=want_feature,
                                          points=pts, radius=radius, eps=eps)
        gs = torch.randn(sdf7t.shape, generator=torch.Generator().manual_seed(1)).to(dev)
        loss = (sdf7t * gs).sum()
        if want_feature:
            gf = torch.randn(feat.shape, generator=torch.Generator().manual_seed(2)).to(dev)
            loss = loss + (feat * gf).sum()
        loss.backward()
        outs[flag] = (sdf7t.detach().clone(), None if feat is None else feat.detach().clone(),
                      [t.grad.clone() for t in [table] + [p for wb in ws for p in wb]])
    a, b = outs["0"], outs["1"]
    sc = float(a[0].abs().max())
    assert float((a[0] - b[0]).abs().max()) < 2e-6 * sc
    if want_feature:
        assert float((a[1] - b[1]).abs().max()) < 2e-6 * float(a[1].abs().max())
    names = ["table", "w0", "b0", "w1", "b1", "w2", "b2"]
    for n, ga, gb in zip(names, a[2], b[2]):
        scale = float(ga.abs().max())
        assert scale > 0 and bool(torch.isfinite(gb).all()), n
        assert float((ga - gb).abs().max()) < 2e-5 * scale, (n, float((ga - gb).abs().max()) / scale)


def test_x3_field_vs_fp64(dev, ops):
    """The x3 forward against an fp64 evaluation of the same network on the fp32 hash features (the accuracy the fp32
    headline rests on: tests/test_gpu_edges.py holds the planes path to the same bar)."""
    from rise_sdf_amd import _lib, fused
    H, N2, S = 64, 13, 3000
    meta, table, ws, x7t, pts, radius, eps = _field_inputs(dev, ops, S, H, N2, seed=9)
    with torch.no_grad():
        sdf7t, feat = fused.sdf_field_fd7(x7t, table, ws, meta, 16, 2.0, -1.0, eps / (2 * radius), want_feature=True,
                                          points=pts, radius=radius, eps=eps)
        planes = torch.zeros(16, 7, S, 2, device=dev)
        assert _lib.lib().rsdf_hashgrid_fwd_fd7_pts(_lib.ptr(pts), radius, eps, _lib.ptr(table), ctypes.byref(meta), S, 16,
                                                    _lib.ptr(planes), _lib.stream_ptr()) == 0
        X = torch.cat([x7t * 2.0 - 1.0, planes.permute(1, 2, 0, 3).reshape(7, S, 32)], dim=-1).double()   # [7,S,35]
        sp = lambda z: torch.nn.functional.softplus(z, beta=100)
        (w0, b0), (w1, b1), (w2, b2) = [(w.double(), b.double()) for w, b in ws]
        out = sp(sp(X @ w0.T + b0) @ w1.T + b1) @ w2.T + b2                                                 # [7,S,N2]
    err = float((sdf7t.double() - out[..., 0]).abs().max())
    scale = float(out[..., 0].abs().max())
    assert err < 2e-6 * scale, (err, scale)
    assert float((feat.double() - out[0]).abs().max()) < 2e-6 * float(out[0].abs().max())
