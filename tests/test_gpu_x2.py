"""The "x2" form of the fused SDF field (round 4: rsdf_hashgrid_fwd_fd7_x2, rsdf_sdfmlp_fd7_fwd_x2 / _bwd_x2, csrc/mlp_x2.hip):
every fp32 matrix operand as two fp16 parts, three matrix instructions per product, the input image pre-split by the gather.

The image must be EXACTLY the two-part split of what the fp32 stencil gather writes (which is bit-exact vs the oracle), and
the MLP kernels that consume it must agree with the round-3 kernels (three bf16 parts, six products) and with fp64 at the
accuracy of an fp32 GEMM chain."""
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


def _split2(x):
    """The kernels' split (mlp_x2.hip split2_pair) of the scaled value: round to nearest even at both steps."""
    xs = x * 256.0
    h = xs.to(torch.float16).to(torch.float32)
    l = (xs - h).to(torch.float16).to(torch.float32)
    return h, l


@pytest.mark.parametrize("parts", [2, 1])
@pytest.mark.parametrize("S,n_active,form", [(5000, 16, "pts"), (4097, 16, "x7t"), (777, 5, "pts"), (31, 16, "pts")])
def test_x2_image_is_the_exact_split_of_the_planes(dev, ops, S, n_active, form, parts):
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
    Sp = int(L.rsdf_x2_rows(S))
    assert Sp % 32 == 0 and 0 <= Sp - S < 32 and int(L.rsdf_x2_bytes(S, parts)) == 7 * parts * 36 * Sp * 2 + 4096
    buf = torch.full((int(L.rsdf_x2_bytes(S, parts)) // 2,), 0x7E00, dtype=torch.int16, device=dev)   # NaN patterns: every slot must be written
    x3 = buf[:Sp // 32 * 7 * parts * 36 * 32].view(Sp // 32, 7, parts, 36, 32)
    assert L.rsdf_hashgrid_fwd_fd7_x2(_lib.ptr(x7t) if form == "x7t" else None, _lib.ptr(pts) if form == "pts" else None,
                                      radius, eps, _lib.ptr(tg), ctypes.byref(meta_g), S, n_active, xyz_scale, xyz_offset,
                                      parts, _lib.ptr(x3), _lib.stream_ptr()) == 0
    torch.cuda.synchronize()
    # [tile][tap][part][column][32 rows], the row halves of columns with bit 3 set swapped -> [7, 3, 36, Sp]
    swapped = ((torch.arange(36, device=dev) >> 3) & 1).bool()
    x3 = torch.where(swapped[None, None, None, :, None], torch.cat([x3[..., 16:], x3[..., :16]], dim=-1), x3)
    vals = x3.view(torch.float16).to(torch.float32).permute(1, 2, 3, 0, 4).reshape(7, parts, 36, Sp)
    assert bool((vals[..., S:] == 0).all()), "rows past n_samples must be zeros"
    assert bool((vals[:, 0, 35, :S] == 256).all()) and bool((vals[:, 1:, 35, :S] == 0).all()), "bias column"
    # hash-feature columns: column 2 l + f of tap t = planes[l, t, :, f]
    want = planes.permute(1, 0, 3, 2).reshape(7, 32, S)                             # [tap, 2 l + f, S]
    h, l = _split2(want)
    got = vals[:, :, :32, :S]
    assert torch.equal(got[:, 0], h)
    if parts == 2:
        assert torch.equal(got[:, 1], l)
        err = ((got[:, 0].double() + got[:, 1].double()) / 256.0 - want.double()).abs()
        # two 11-bit roundings: one fp32 ulp of the value (2^-23 relative), 2^-25 / 256 absolute below the fp16 normal range
        assert bool((err <= want.double().abs() * 2.0 ** -23 + 2.0 ** -33).all()), "hi + lo must give the value back to one fp32 ulp"
    if n_active < 16:
        assert bool((got[:, :, 2 * n_active:] == 0).all())
    # xyz columns: the tap's unit-cube coordinates (what rsdf_fd_points wrote) * scale + offset
    xyz = (x7t * xyz_scale + xyz_offset).permute(0, 2, 1)                           # [7, 3, S]
    hx, lx = _split2(xyz)
    gx = vals[:, :, 32:35, :S]
    assert torch.equal(gx[:, 0], hx) and (parts == 1 or torch.equal(gx[:, 1], lx))


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
    ws = [(mk(H, K0, sc=0.4), mk(H, sc=0.1)), (mk(H, H, sc=2.0 / H ** 0.5), mk(H, sc=0.1)), (mk(N2, H, sc=2.4 / H ** 0.5), mk(N2, sc=0.1))]
    x7t, pts = _stencil_points(dev, ops, S, eps, radius, seed=seed + 10)
    return meta_g, table, ws, x7t, pts, radius, eps


@pytest.mark.parametrize("H", [64, 128])
@pytest.mark.parametrize("S,n_active,want_feature", [(4133, 16, True), (2048, 16, False), (1000, 7, True)])
def test_x2_field_matches_the_round3_kernels(dev, ops, S, n_active, want_feature, H, monkeypatch):
    """rise_sdf_amd.fused.sdf_field_fd7 at H = 64 in the x2 form against the round-3 kernels (three bf16 parts, six
    products, fp32 planes) on the same inputs: values to 2e-6 of the largest, gradients to 3e-5."""
    from rise_sdf_amd import fused
    N2 = 13
    meta, table, ws, x7t, pts, radius, eps = _field_inputs(dev, ops, S, H, N2)
    eps_unit = eps / (2 * radius)
    outs = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("RSDF_X2", flag)
        assert fused.use_x2(35, H, N2, "fp32") == (flag == "1")
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
        assert float((ga - gb).abs().max()) < 3e-5 * scale, (n, float((ga - gb).abs().max()) / scale)


@pytest.mark.parametrize("H", [64, 128])
def test_x2_field_vs_fp64(dev, ops, H, monkeypatch):
    """The x2 forward against an fp64 evaluation of the same network on the fp32 hash features, next to the round-3 kernels
    and to torch's fp32 GEMM chain on the same inputs: the x2 form must be as accurate as an fp32 GEMM chain."""
    from rise_sdf_amd import _lib, fused
    N2, S = 13, 20000
    meta, table, ws, x7t, pts, radius, eps = _field_inputs(dev, ops, S, H, N2, seed=9)
    res = {}
    with torch.no_grad():
        for flag in ("1", "0"):
            monkeypatch.setenv("RSDF_X2", flag)
            res[flag] = fused.sdf_field_fd7(x7t, table, ws, meta, 16, 2.0, -1.0, eps / (2 * radius), want_feature=True,
                                            points=pts, radius=radius, eps=eps)
        planes = torch.zeros(16, 7, S, 2, device=dev)
        assert _lib.lib().rsdf_hashgrid_fwd_fd7_pts(_lib.ptr(pts), radius, eps, _lib.ptr(table), ctypes.byref(meta), S, 16,
                                                    _lib.ptr(planes), _lib.stream_ptr()) == 0
        X32 = torch.cat([x7t * 2.0 - 1.0, planes.permute(1, 2, 0, 3).reshape(7, S, 32)], dim=-1)            # [7,S,35]
        sp = lambda z: torch.nn.functional.softplus(z, beta=100)
        chain = lambda X, W: sp(sp(X @ W[0][0].T + W[0][1]) @ W[1][0].T + W[1][1]) @ W[2][0].T + W[2][1]
        out = chain(X32.double(), [(w.double(), b.double()) for w, b in ws])                                # [7,S,N2]
        t32 = chain(X32, ws).double()
    scale = float(out[..., 0].abs().max())
    e_x2 = float((res["1"][0].double() - out[..., 0]).abs().max()) / scale
    e_r3 = float((res["0"][0].double() - out[..., 0]).abs().max()) / scale
    e_t32 = float((t32[..., 0] - out[..., 0]).abs().max()) / scale
    # the finite-difference normal divides an SDF difference by 2 eps: what the stencil's error does to it
    fd = lambda v: (v[1] - v[2]) / (2 * eps)
    n_x2 = float((fd(res["1"][0].double()) - fd(out[..., 0])).abs().max())
    n_t32 = float((fd(t32[..., 0]) - fd(out[..., 0])).abs().max())
    print(f"H = {H}: max SDF error / max|sdf|: x2 {e_x2:.2e}, round-3 kernels {e_r3:.2e}, torch fp32 chain {e_t32:.2e}; "
          f"FD normal component error: x2 {n_x2:.2e}, torch fp32 {n_t32:.2e}")
    assert e_x2 < max(1.5 * e_t32, 3e-7), (e_x2, e_t32)
    fs = float(out[0].abs().max())
    assert float((res["1"][1].double() - out[0]).abs().max()) < max(1.5 * float((t32[0] - out[0]).abs().max()), 3e-7 * fs)


@pytest.mark.parametrize("H", [64, 128])
def test_x1_field_is_the_fp16_operand_network(dev, ops, H):
    """precision 'fp16' (ONE fp16 part: the 16-bit mode of the fused SDF field, BASELINE.json configs[4]) against the oracle's
    network with every matrix operand rounded once to fp16 (``oracle.mlp_precision("fp16")``), forward and backward: it must
    be that network (closer to it than to the fp32 one), and an order of magnitude closer to fp32 than the bf16 mode is."""
    from rise_sdf_amd import _lib, fused
    N2, S = 13, 6000
    meta, table, ws, x7t, pts, radius, eps = _field_inputs(dev, ops, S, H, N2, seed=21)
    eps_unit = eps / (2 * radius)
    gs = torch.randn(7, S, generator=torch.Generator().manual_seed(1)).to(dev)
    gf = torch.randn(S, N2, generator=torch.Generator().manual_seed(2)).to(dev)

    def run_hip(prec):
        for t in [table] + [p for wb in ws for p in wb]:
            t.grad = None
        sdf7t, feat = fused.sdf_field_fd7(x7t, table, ws, meta, 16, 2.0, -1.0, eps_unit, want_feature=True, points=pts,
                                          radius=radius, eps=eps, precision=prec)
        ((sdf7t * gs).sum() + (feat * gf).sum()).backward()
        return sdf7t.detach().clone(), feat.detach().clone(), [t.grad.clone() for t in [table] + [p for wb in ws for p in wb]]

    def run_oracle(prec):
        planes = torch.zeros(16, 7, S, 2, device=dev)
        assert _lib.lib().rsdf_hashgrid_fwd_fd7_pts(_lib.ptr(pts), radius, eps, _lib.ptr(table), ctypes.byref(meta), S, 16,
                                                    _lib.ptr(planes), _lib.stream_ptr()) == 0
        X = torch.cat([planes.permute(1, 2, 0, 3).reshape(7, S, 32), x7t * 2.0 - 1.0], dim=-1).cpu().reshape(7 * S, 35)
        ps = [{"w": torch.cat([w[:, 3:], w[:, :3]], dim=1).detach().cpu().clone().requires_grad_(True) if i == 0
               else w.detach().cpu().clone().requires_grad_(True), "b": b.detach().cpu().clone().requires_grad_(True)}
              for i, (w, b) in enumerate(ws)]
        with oracle.mlp_precision(prec):
            out = oracle.vanilla_mlp(X, ps).view(7, S, N2)
            ((out[..., 0] * gs.cpu()).sum() + (out[0] * gf.cpu()).sum()).backward()
        return out[..., 0].detach(), out[0].detach(), ps

    s16, f16, g16 = run_hip("fp16")
    s32, f32, g32 = run_hip("fp32")
    o16, of16, p16 = run_oracle("fp16")
    o32, _, _ = run_oracle("fp32")
    scale = float(o32.abs().max())
    assert float((s32.cpu() - o32).abs().max()) < 2e-6 * scale
    d_mode = float((s16.cpu() - o32).abs().max()) / scale              # the mode is in force: ~ an fp16 epsilon away from fp32
    assert 2e-5 < d_mode < 5e-3, d_mode
    # ... and it is the fp16-operand network up to rounding-boundary flips: two evaluations that round the same operands differ
    # where the fp32 accumulation order moves an activation across an fp16 rounding boundary (2^-11 of it; tests/test_gpu_bf16.py
    # holds the bf16 mode to the same kind of bar, 4e-3 = one bf16 epsilon)
    err = float((s16.cpu() - o16).abs().max()) / scale
    assert err < 2e-3, (err, d_mode)
    assert float((f16.cpu() - of16).abs().max()) < 2e-3 * float(of16.abs().max())
    cos = lambda a, b: float((a.double().flatten() * b.double().flatten()).sum() / (a.double().norm() * b.double().norm() + 1e-300))
    w0_o = torch.cat([p16[0]["w"].grad[:, 32:], p16[0]["w"].grad[:, :32]], dim=1)           # back to [xyz | features]
    for name, got, want in (("w0", g16[1], w0_o), ("b0", g16[2], p16[0]["b"].grad), ("w1", g16[3], p16[1]["w"].grad),
                            ("b1", g16[4], p16[1]["b"].grad), ("w2", g16[5], p16[2]["w"].grad), ("b2", g16[6], p16[2]["b"].grad)):
        assert cos(got.cpu(), want) > 0.9999, (name, cos(got.cpu(), want))
        assert float((got.cpu() - want).abs().max()) < 1e-2 * float(want.abs().max()), name
    assert cos(g16[0], g32[0]) > 0.9999                                 # table gradient: against the fp32 kernels
    sb, _, _ = run_hip("bf16")
    d_bf16 = float((sb.cpu() - o32).abs().max()) / scale
    print(f"H = {H}: |sdf - fp32| / max|sdf|: fp16 mode {d_mode:.2e}, bf16 mode {d_bf16:.2e}; fp16 vs its oracle {err:.2e}")
    assert d_mode < 0.35 * d_bf16


def test_x2_range_preconditions(dev, ops, monkeypatch):
    """The x2 form's fp16 class scales bound |weight| < 1023, |input| < 255 and |activation| < 454.  Inside the bounds large
    values are exact citizens (a weight of 900 -- without the feature rows, whose product would split an activation of ~600 --
    gives the range-free kernels' result); outside them the result is inf / nan -- never a
    silently wrong finite number -- and RSDF_CHECK=1 names the violation up front."""
    from rise_sdf_amd import _lib, fused
    monkeypatch.delenv("RSDF_CHECK", raising=False)          # (the suite may itself run under RSDF_CHECK=1)
    H, N2, S = 64, 13, 1000
    meta, table, ws, x7t, pts, radius, eps = _field_inputs(dev, ops, S, H, N2, seed=33)
    with torch.no_grad():
        ws[1][0][3, 5] = 900.0                               # one large hidden weight, still in range
        sdf_ok, _ = fused.sdf_field_fd7(x7t, table, ws, meta, 16, 2.0, -1.0, eps / (2 * radius), points=pts, radius=radius, eps=eps)
        monkeypatch.setenv("RSDF_X2", "0")
        sdf_r3, _ = fused.sdf_field_fd7(x7t, table, ws, meta, 16, 2.0, -1.0, eps / (2 * radius), points=pts, radius=radius, eps=eps)
        monkeypatch.setenv("RSDF_X2", "1")
        assert bool(torch.isfinite(sdf_ok).all())
        assert float((sdf_ok - sdf_r3).abs().max()) < 1e-5 * float(sdf_r3.abs().max())
        ws[1][0][3, 5] = 5000.0                              # out of range: overflows to inf / nan, visibly
        sdf_bad, _ = fused.sdf_field_fd7(x7t, table, ws, meta, 16, 2.0, -1.0, eps / (2 * radius), points=pts, radius=radius, eps=eps)
        assert not bool(torch.isfinite(sdf_bad).all())
        monkeypatch.setenv("RSDF_CHECK", "1")
        with pytest.raises(_lib.RiseSdfHipError, match="RSDF_CHECK x2"):
            fused.sdf_field_fd7(x7t, table, ws, meta, 16, 2.0, -1.0, eps / (2 * radius), points=pts, radius=radius, eps=eps)
