"""The layer-pair kernels of the 128-wide radiance networks (csrc/mlp_pair.hip: rsdf_pair_pack / _fwd / _bwd; models/texture.py:237-327)
against the per-layer kernels (RSDF_PAIR=0: split-bf16 products, one kernel per layer) and against an fp64 evaluation of the
same network: the pair path must be as accurate as an fp32 GEMM chain, forward and backward."""
import pytest
import torch

import oracle  # noqa: F401  (conftest path)

pytestmark = pytest.mark.gpu


def _net(dev, K, nh, N2, seed):
    g = torch.Generator().manual_seed(seed)
    dims = [K] + [128] * nh + [N2]
    layers = []
    for i in range(len(dims) - 1):
        bound = (6.0 / dims[i]) ** 0.5                    # kaiming_uniform_(relu), as the reference initialises these networks
        w = ((torch.rand(dims[i + 1], dims[i], generator=g) * 2 - 1) * bound).to(dev).requires_grad_(True)
        b = ((torch.rand(dims[i + 1], generator=g) * 2 - 1) * 0.1).to(dev).requires_grad_(True)
        layers.append((w, b))
    return layers


def _run(ops, x, layers, acts, go, dx_cols=None):
    for w, b in layers:
        w.grad = b.grad = None
    x = x.detach().clone().requires_grad_(x.requires_grad)
    y = ops.mlp_chain(x, layers, acts, dx_cols=dx_cols)
    (y * go).sum().backward()
    return y.detach(), x.grad, [t.grad.clone() for wb in layers for t in wb]


def test_pair_image_round_trip(dev):
    from rise_sdf_amd import _lib
    L = _lib.lib()
    n, K = 1000, 84
    x = (torch.randn(n, K, generator=torch.Generator().manual_seed(0)) * 3).to(dev)
    img = torch.full((int(L.rsdf_pair_image_bytes(n)),), 0x7F, dtype=torch.uint8, device=dev)
    rows = torch.empty(n, 128, device=dev)
    assert L.rsdf_pair_pack(_lib.ptr(x), K, K, n, _lib.ptr(img), None, _lib.stream_ptr()) == 0
    assert L.rsdf_pair_unpack(_lib.ptr(img), n, _lib.ptr(rows), _lib.stream_ptr()) == 0
    torch.cuda.synchronize()
    # hi + lo reproduces the value to one fp32 ulp (two 11-bit roundings of the x 2^6 value); columns >= K are zeros
    assert float((rows[:, :K] - x).abs().max()) <= float(x.abs().max()) * 2.0 ** -23
    assert bool((rows[:, K:96] == 0).all())          # (chunks past the last 32-column group that holds a column are not written)


@pytest.mark.parametrize("K,nh,N2,n,out_act", [(84, 4, 6, 4133, "sigmoid"), (73, 4, 3, 1000, "sigmoid"), (84, 2, 1, 2048, "sigmoid"),
                                               (84, 2, 2, 31, "none"), (128, 4, 3, 777, "none"), (17, 2, 5, 65, "none"),
                                               (76, 4, 3, 33, "none"), (84, 2, 13, 500, "none"), (84, 4, 8, 700, "sigmoid")])
def test_pair_chain_matches_per_layer_kernels_and_fp64(dev, K, nh, N2, n, out_act, monkeypatch):
    from rise_sdf_amd import ops
    layers = _net(dev, K, nh, N2, seed=K + nh)
    acts = ["relu"] * nh + [out_act]
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, K, generator=g).to(dev).requires_grad_(True)
    go = torch.randn(n, N2, generator=g).to(dev)
    monkeypatch.setenv("RSDF_PAIR", "0")
    ref = _run(ops, x, layers, acts, go)
    monkeypatch.setenv("RSDF_PAIR", "1")
    assert ops.pair_chain_ok(x, [w for w, _ in layers], [b for _, b in layers], tuple(ops.L.ACT_IDS[a] for a in acts), "fp32")
    got = _run(ops, x, layers, acts, go)
    # fp64 evaluation of the same network
    l64 = [(w.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)) for w, b in layers]
    x64 = x.detach().double().requires_grad_(True)
    h = x64
    for i, (w, b) in enumerate(l64):
        h = h @ w.T + b
        h = torch.relu(h) if i < nh else (torch.sigmoid(h) if out_act == "sigmoid" else h)
    (h * go.double()).sum().backward()
    g64 = [t.grad for wb in l64 for t in wb]
    sc = float(h.abs().max())
    e_pair, e_layer = float((got[0].double() - h).abs().max()) / sc, float((ref[0].double() - h).abs().max()) / sc
    print(f"K {K} nh {nh} N2 {N2} n {n}: output error / max: pair {e_pair:.1e}, per-layer {e_layer:.1e}")
    assert e_pair < max(3e-7 * nh, 3 * e_layer)
    names = [f"{'wb'[j]}{i}" for i in range(nh + 1) for j in range(2)]
    worst = 0.0
    for name, a, b, c in zip(["dx"] + names, [got[1]] + got[2], [ref[1]] + ref[2], [x64.grad] + g64):
        s = float(c.abs().max())
        ea, eb = float((a.double() - c).abs().max()) / s, float((b.double() - c).abs().max()) / s
        worst = max(worst, ea)
        assert ea < max(2e-6, 4 * eb), (name, ea, eb)
    print(f"   gradients vs fp64, worst tensor: {worst:.1e}")


@pytest.mark.parametrize("N2,scale", [(1, 0.05), (1, 1e-4), (2, 0.02), (6, 2.0)])
def test_pair_backward_output_layer_of_any_magnitude(dev, N2, scale, monkeypatch):
    """The folded dW_out product scales its dz_out image by max|dz_out|, not by the bound on |d hb| (= max|dz_out| x the largest
    column sum of |W_out|, BELOW max|dz_out| for small output weights): a fresh 1 x 128 roughness layer (models/texture.py:
    316-319, torch's default Linear init: |w| <= 0.088) sent dz_out x scale past fp16 and dW_out to NaN after one step."""
    from rise_sdf_amd import ops
    layers = _net(dev, 84, 2, N2, seed=3)
    with torch.no_grad():
        layers[-1][0].mul_(scale / 0.2)
    acts = ["relu", "relu", "sigmoid"]
    g = torch.Generator().manual_seed(8)
    x = torch.randn(3000, 84, generator=g).to(dev).requires_grad_(True)
    go = torch.randn(3000, N2, generator=g).to(dev)
    monkeypatch.setenv("RSDF_PAIR", "0")
    ref = _run(ops, x, layers, acts, go)
    monkeypatch.setenv("RSDF_PAIR", "1")
    got = _run(ops, x, layers, acts, go)
    for name, a, b in zip(["y", "dx"] + [f"{'wb'[j]}{i}" for i in range(3) for j in range(2)], [got[0], got[1]] + got[2],
                          [ref[0], ref[1]] + ref[2]):
        assert bool(torch.isfinite(a).all()), name
        # (y: a sigmoid of logits up to +-20 x the output scale turns 2e-7 of the largest logit into up to 1e-5 of the output)
        tol = 1e-5 if name == "y" else 3e-6
        assert float((a - b).abs().max()) <= tol * float(b.abs().max()) + 1e-30, (name, float((a - b).abs().max()), float(b.abs().max()))


@pytest.mark.parametrize("N2,out_act", [(6, "sigmoid"), (1, "none"), (8, "sigmoid")])
def test_pair_forward_output_layer_fold(dev, N2, out_act, monkeypatch):
    """The narrow output layer computed inside the last pair's forward kernel (two waves, hb from its LDS image) against the
    per-layer kernel on the same h_last (RSDF_PAIR_FOLD_FWD=0)."""
    from rise_sdf_amd import ops
    layers = _net(dev, 84, 4, N2, seed=31)
    acts = ["relu"] * 4 + [out_act]
    x = torch.randn(3000, 84, generator=torch.Generator().manual_seed(2)).to(dev)
    with torch.no_grad():
        monkeypatch.setenv("RSDF_PAIR_FOLD_FWD", "0")
        a = ops.mlp_chain(x, layers, acts)
        monkeypatch.setenv("RSDF_PAIR_FOLD_FWD", "1")
        b = ops.mlp_chain(x, layers, acts)
        c = ops.mlp_chain(x, layers, acts)
    assert torch.equal(b, c), "the folded output layer must be deterministic"
    assert float((a - b).abs().max()) < 2e-6 * max(float(a.abs().max()), 1.0)


@pytest.mark.parametrize("nh,K2", [(4, 36), (2, 25), (4, 25)])
def test_pair_chain_two_source_input(dev, monkeypatch, nh, K2):
    """mlp_chain(x, ..., x2=...): the network's input cat([x, x2], -1) is packed from its two sources (rsdf_pair_pack2) and the
    input gradient comes back per source; an x2 that needs no gradient narrows the dx window to x's columns."""
    from rise_sdf_amd import ops
    layers = _net(dev, 48 + K2, nh, 6, seed=41)
    acts = ["relu"] * nh + ["sigmoid"]
    g = torch.Generator().manual_seed(3)
    a = torch.randn(2001, 48, generator=g).to(dev).requires_grad_(True)
    go = torch.randn(2001, 6, generator=g).to(dev)
    for b_grad in (True, False):
        # (both sources with gradients: the kernel writes them as two contiguous tensors, the second with a row length that
        #  is not a multiple of four for K2 = 25: the SH(5) encoding of the specular network)
        b = torch.randn(2001, K2, generator=torch.Generator().manual_seed(4)).to(dev).requires_grad_(b_grad)
        res = []
        for two in (False, True):
            for t in [a, b] + [p for wb in layers for p in wb]:
                t.grad = None
            y = ops.mlp_chain(a, layers, acts, x2=b) if two else ops.mlp_chain(torch.cat([a, b], -1), layers, acts)
            (y * go).sum().backward()
            res.append((y.detach().clone(), a.grad.clone(), None if b.grad is None else b.grad.clone(),
                        [p.grad.clone() for wb in layers for p in wb]))
        (y0, ga0, gb0, gw0), (y1, ga1, gb1, gw1) = res
        assert torch.equal(y0, y1)
        assert float((ga0 - ga1).abs().max()) < 1e-6 * float(ga0.abs().max())
        assert (gb0 is None) == (gb1 is None) and (gb0 is None or float((gb0 - gb1).abs().max()) < 1e-6 * float(gb0.abs().max()))
        for p0, p1 in zip(gw0, gw1):
            assert float((p0 - p1).abs().max()) < 1e-6 * float(p0.abs().max())


def test_pair_backward_mask_sources_agree(dev):
    """rsdf_pair_bwd takes the ReLU mask of the pair's upper layer three ways: recomputed from x (one more product), from the
    forward's own hb rows, or already applied to g by the producer.  Same dx and weight gradients."""
    from rise_sdf_amd import _lib
    L = _lib.lib()
    p, st = _lib.ptr, _lib.stream_ptr()
    n, K = 3000, 84
    (wa, ba), (wb, bb) = [(w.detach().contiguous(), b.detach().contiguous()) for w, b in _net(dev, K, 2, 1, seed=21)[:2]]
    g = torch.Generator().manual_seed(8)
    x = torch.randn(n, K, generator=g).to(dev)
    gy = torch.randn(n, 128, generator=g).to(dev)
    img = torch.empty(int(L.rsdf_pair_image_bytes(n)), dtype=torch.uint8, device=dev)
    hb = torch.empty(n, 128, device=dev)
    assert L.rsdf_pair_pack(p(x), K, K, n, p(img), None, st) == 0
    assert L.rsdf_pair_fwd(p(img), K, p(wa), p(ba), p(wb), p(bb), n, None, p(hb), None, None, 0, 0, None, None, st) == 0
    ref = torch.relu(torch.relu(x.double() @ wa.double().T + ba.double()) @ wb.double().T + bb.double())
    assert float((hb.double() - ref).abs().max()) < 3e-7 * float(ref.abs().max())
    bound = torch.zeros(2, dtype=torch.int32, device=dev)
    assert L.rsdf_pair_bound_from_rows(p(gy), gy.numel(), p(bound), st) == 0
    outs = []
    for mode in ("recompute", "hb_rows", "premasked"):
        dx = torch.full((n, K), float("nan"), device=dev)
        gr = [torch.zeros_like(t) for t in (wa, ba, wb, bb)]
        gin = gy * (hb > 0) if mode == "premasked" else gy
        _lib.check(L.rsdf_pair_bwd(p(img), K, p(wa), p(ba), p(wb), p(bb), n, p(gin.contiguous()), int(mode == "premasked"),
                                   p(hb) if mode == "hb_rows" else None, None, None, 0, None, p(bound), p(dx), K, K, None, 0, 0, 0, None,
                                   *[p(t) for t in gr], st),
                   "pair_bwd")
        torch.cuda.synchronize()
        outs.append([dx] + gr)
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert float((a - b).abs().max()) < 2e-6 * float(a.abs().max())


def test_pair_chain_input_window_and_frozen_input(dev, monkeypatch):
    """dx_cols (only a window of the input columns needs a gradient) and an input that needs none."""
    from rise_sdf_amd import ops
    K, nh, N2, n = 84, 4, 6, 1500
    layers = _net(dev, K, nh, N2, seed=3)
    acts = ["relu"] * nh + ["none"]
    g = torch.Generator().manual_seed(6)
    x = torch.randn(n, K, generator=g).to(dev).requires_grad_(True)
    go = torch.randn(n, N2, generator=g).to(dev)
    res = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("RSDF_PAIR", flag)
        res[flag] = _run(ops, x, layers, acts, go, dx_cols=(0, 48)), _run(ops, x.detach(), layers, acts, go)
    (a_win, a_noin), (b_win, b_noin) = res["0"], res["1"]
    assert b_noin[1] is None and bool((b_win[1][:, 48:] == 0).all())
    assert float((a_win[1] - b_win[1]).abs().max()) < 3e-5 * float(a_win[1].abs().max())
    for ga, gb in zip(a_noin[2], b_noin[2]):
        assert float((ga - gb).abs().max()) < 3e-5 * float(ga.abs().max())


def test_pair_forward_range_violation_is_counted(dev):
    """An operand beyond the fp16 class range (|input|, |weight| or |hidden activation| >= 1023) is counted in the device's status
    words at the point where it would be split, and the next check raises the named error."""
    import rise_sdf_amd as R
    from rise_sdf_amd import _lib, ops
    layers = _net(dev, 84, 2, 3, seed=9)
    with torch.no_grad():
        layers[0][1][5] = 3000.0                             # a bias that drives one hidden activation past 1023
        x = torch.randn(500, 84, generator=torch.Generator().manual_seed(1)).to(dev)
        R.check_status(dev)
        ops.mlp_chain(x, layers, ["relu", "relu", "none"])
        # (ReLU's max() swallows the NaN that the overflowed operand makes downstream: the outputs may well be finite, which is
        # why the guard sits at the split points and not at the outputs)
        with pytest.raises(_lib.RiseSdfHipError, match="RSDF_PAIR=0"):
            R.check_status(dev)
        layers[0][1][5] = 0.1
        layers[1][0][7, 9] = 2000.0                          # a weight beyond the class range
        ops.mlp_chain(x, layers, ["relu", "relu", "none"])
        with pytest.raises(_lib.RiseSdfHipError, match="RSDF_PAIR=0"):
            R.check_status(dev)
        layers[1][0][7, 9] = 0.1
        ops.mlp_chain(x * 2000.0, layers, ["relu", "relu", "none"])      # inputs beyond it
        with pytest.raises(_lib.RiseSdfHipError, match="RSDF_PAIR=0"):
            R.check_status(dev)
        ops.mlp_chain(x, layers, ["relu", "relu", "none"])
        assert R.check_status(dev)["x2_fwd_nonfinite"] == 0
