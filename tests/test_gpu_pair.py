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


@pytest.mark.parametrize("K1,K2,ld1,ld2,off1,off2", [(48, 39, 48, 39, 0, 0),    # the material networks' [feature | xyz encoding]
                                                       (48, 16, 48, 16, 0, 0),    # [feature | SH]: both sources as float4
                                                       (48, 16, 52, 20, 4, 4),    # rows wider than their columns, aligned windows
                                                       (48, 16, 49, 17, 1, 1),    # unaligned bases and strides: the scalar path
                                                       (44, 20, 44, 20, 0, 0),    # K1 not a multiple of 8: a chunk spans both sources
                                                       (128, 0, 128, 0, 0, 0)])
def test_pair_pack_two_sources_every_alignment(dev, K1, K2, ld1, ld2, off1, off2):
    """rsdf_pair_pack2 reads aligned sources as two float4 per 8-column chunk (round 6) and everything else column by column:
    both paths must write the same image as the values themselves (pack -> unpack round trip, 2^-22 relative)."""
    from rise_sdf_amd import _lib
    L = _lib.lib()
    n = 1003
    g = torch.Generator().manual_seed(K1 * 131 + ld1)
    b1 = (torch.randn(n, ld1 + off1, generator=g) * 3).to(dev)
    b2 = (torch.randn(n, max(ld2 + off2, 1), generator=g) * 3).to(dev)
    x1 = b1.view(-1)[off1:off1 + (n - 1) * ld1 + K1]            # a window at column off1 of the first row: row stride ld1
    x2 = b2.view(-1)[off2:off2 + (n - 1) * ld2 + K2] if K2 else None
    img = torch.full((int(L.rsdf_pair_image_bytes(n)),), 0x7F, dtype=torch.uint8, device=dev)
    rows = torch.empty(n, 128, device=dev)
    assert L.rsdf_pair_pack2(_lib.ptr(x1), ld1, K1, _lib.ptr(x2) if K2 else None, ld2, K2, n, _lib.ptr(img), None,
                             _lib.stream_ptr()) == 0
    assert L.rsdf_pair_unpack(_lib.ptr(img), n, _lib.ptr(rows), _lib.stream_ptr()) == 0
    want = torch.zeros(n, 128, device=dev)
    want[:, :K1] = torch.as_strided(x1, (n, K1), (ld1, 1))
    if K2:
        want[:, K1:K1 + K2] = torch.as_strided(x2, (n, K2), (ld2, 1))
    kw = ((K1 + K2 + 31) // 32) * 32        # (chunks past the last 32-column group that holds a column are not written)
    assert float((rows[:, :kw] - want[:, :kw]).abs().max()) <= 2.0 ** -21 * float(want.abs().max())
    assert bool((rows[:, K1 + K2:kw] == 0).all())


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


def test_pair_forward_range_violation_is_counted(dev, monkeypatch):
    """An operand beyond the fp16 class range (|input|, |weight| or |hidden activation| >= 1023) is counted in the device's status
    words at the point where it would be split.  Default policy: the next check switches the radiance networks to the per-layer
    kernels (fp32's range) and ``rise_sdf_amd.guarded`` hands back what RSDF_PAIR=0 computes; RSDF_RANGE_ERROR=raise: the
    named error."""
    import warnings
    import rise_sdf_amd as R
    from rise_sdf_amd import _lib, ops
    layers = _net(dev, 84, 2, 3, seed=9)
    acts = ["relu", "relu", "none"]
    with torch.no_grad():
        x = torch.randn(500, 84, generator=torch.Generator().manual_seed(1)).to(dev)
        monkeypatch.setenv("RSDF_RANGE_ERROR", "raise")
        layers[0][1][5] = 3000.0                             # a bias that drives one hidden activation past 1023
        R.check_status(dev)
        ops.mlp_chain(x, layers, acts)
        # (ReLU's max() swallows the NaN that the overflowed operand makes downstream: the outputs may well be finite, which is
        # why the guard sits at the split points and not at the outputs)
        with pytest.raises(_lib.RiseSdfHipError, match="RSDF_PAIR=0"):
            R.check_status(dev)
        layers[0][1][5] = 0.1
        layers[1][0][7, 9] = 2000.0                          # a weight beyond the class range
        ops.mlp_chain(x, layers, acts)
        with pytest.raises(_lib.RiseSdfHipError, match="RSDF_PAIR=0"):
            R.check_status(dev)
        layers[1][0][7, 9] = 0.1
        ops.mlp_chain(x * 2000.0, layers, acts)      # inputs beyond it
        with pytest.raises(_lib.RiseSdfHipError, match="RSDF_PAIR=0"):
            R.check_status(dev)
        ops.mlp_chain(x, layers, acts)
        assert R.check_status(dev)["x2_fwd_nonfinite"] == 0
        # ---- default policy: reroute
        monkeypatch.delenv("RSDF_RANGE_ERROR")
        layers[0][1][5] = 3000.0
        monkeypatch.setenv("RSDF_PAIR", "0")
        ref = ops.mlp_chain(x, layers, acts)
        monkeypatch.delenv("RSDF_PAIR")
        assert not _lib.range_free("pair")
        with warnings.catch_warnings(record=True) as wlist:
            warnings.simplefilter("always")
            got = R.guarded(lambda: ops.mlp_chain(x, layers, acts), dev)
        assert len(wlist) == 1 and "RSDF_PAIR=0" in str(wlist[0].message)
        assert _lib.range_free("pair") and not _lib.range_free("x2")
        assert torch.equal(got, ref) and bool(torch.isfinite(got).all())
        # the fp64 chain: the rerouted values are ordinary fp32 results of the reference's network
        h = x.double()
        for i, (w, b) in enumerate(layers):
            h = h @ w.double().T + b.double()
            h = torch.relu(h) if i < 2 else h
        assert float((got.double() - h).abs().max()) < 1e-5 * float(h.abs().max())
    _lib.reset_range_free()


def test_training_step_skips_the_optimizer_when_the_render_pass_overflows(dev, monkeypatch):
    """the radiance networks run AFTER the step's last host read; a range violation there used to reach Adam as
    NaN gradients before any poll.  TrainStep now hands the status word's movement to fused Adam as ``found_inf`` on the
    device: the offending step leaves parameters and moments bit-identical, the next step's sampler read switches the pair
    kernels off, and training continues with finite parameters."""
    import warnings
    from rise_sdf_amd import _lib
    from rise_sdf_amd.step import build_synthetic_training
    monkeypatch.delenv("RSDF_RANGE_ERROR", raising=False)
    _lib.reset_range_free()
    # (128-wide: the radiance networks run on the layer-pair kernels)
    model, ts = build_synthetic_training(dev, stage=0, hidden=128, views=3, res=48, indirect=False, curvature=False,
                                         model_overrides={"train_num_rays": 128, "max_train_num_rays": 256})
    ts.step(0)
    ts.step(1)
    assert float(ts.last_found_inf) == 0.0
    with torch.no_grad():
        bias = model.texture.albedo_network.layers[0].bias
        keep = bias[5].clone()
        bias[5] = 5000.0                                  # a hidden activation past 1023 in the albedo network's first pair
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    moments = [v["exp_avg"].clone() for v in ts.opt.state.values() if "exp_avg" in v]
    ts.step(2)
    assert float(ts.last_found_inf) == 1.0
    for n, p in model.named_parameters():
        assert torch.equal(p.detach(), before[n]), n       # the step was skipped on the device
    for a, b in zip(moments, [v["exp_avg"] for v in ts.opt.state.values() if "exp_avg" in v]):
        assert torch.equal(a, b)
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        ts.step(3)                                        # its sampler's host read sees the count: pair kernels off
    assert any("RSDF_PAIR=0" in str(w.message) for w in wlist)
    assert _lib.range_free("pair")
    assert float(ts.last_found_inf) == 0.0
    ts.step(4)
    moved = sum(int(not torch.equal(p.detach(), before[n])) for n, p in model.named_parameters())
    assert moved > 10
    assert all(bool(torch.isfinite(p).all()) for p in model.parameters())
    with torch.no_grad():
        bias[5] = keep
    _lib.reset_range_free()


@pytest.mark.parametrize("K,nh,N2,out_act,dense", [(84, 4, 6, "sigmoid", False), (84, 4, 6, "sigmoid", True),
                                                   (73, 4, 3, "sigmoid", False), (84, 2, 1, "sigmoid", False),
                                                   (84, 2, 2, "none", True)])
def test_pair_backward_with_the_gradient_spread_of_a_render(dev, K, nh, N2, out_act, dense, monkeypatch):
    """the pair backward shares ONE power-of-two scale per launch (a bound, max|dz_out| x column sums), and rows
    more than ~2^38 below it are flushed; every other test feeds ``randn`` cotangents of uniform magnitude.  Here the
    cotangent rows are scaled by what multiplies them in a render -- the composite weights w_i = alpha_i * prod(1 - alpha_j)
    of marched rays: a *pruned* ray (weights 1e-4 ... 1, its invisible tail removed by the T >= 1e-4 filter) or a *dense*
    c2 ray (no pruning: weights decay to 1e-8 and to exactly 0 behind the surface).  dW / db / dx against fp64 at 1e-5 of each
    tensor's largest entry; the rows that carry the gradient dominate every tensor, which is why a shared scale is enough."""
    from rise_sdf_amd import ops
    n_rays, per_ray = 96, 64
    n = n_rays * per_ray
    g = torch.Generator().manual_seed(11 + K + nh)
    # a NeuS-like ray: alpha ramps up through the surface crossing; late training makes the ramp a step
    t = torch.linspace(-1, 1, per_ray)[None, :] + 0.3 * (torch.rand(n_rays, 1, generator=g) - 0.5)
    sharp = 10.0 ** (1.0 + 2.0 * torch.rand(n_rays, 1, generator=g))
    alpha = torch.sigmoid(t * sharp).clamp(0, 1) * (0.02 + 0.98 * torch.rand(n_rays, 1, generator=g))
    T = torch.cumprod(torch.cat([torch.ones(n_rays, 1), 1 - alpha[:, :-1]], 1), 1)
    w = alpha * T
    if not dense:
        w = torch.where(T >= 1e-4, w, torch.zeros_like(w))          # visibility pruning: those samples are not even there
    w = w.reshape(-1)
    if not dense:
        keep = w > 0
    else:
        keep = torch.ones_like(w, dtype=torch.bool)
    w = w[keep]
    n = int(w.numel())
    spread = float(w[w > 0].min() / w.max())
    assert (spread < 1e-7 or bool((w == 0).any())) if dense else spread < 1e-2
    layers = _net(dev, K, nh, N2, seed=K + nh + 1)
    acts = ["relu"] * nh + [out_act]
    x = torch.randn(n, K, generator=g).to(dev).requires_grad_(True)
    go = (torch.randn(n, N2, generator=g) * w[:, None]).to(dev)
    monkeypatch.setenv("RSDF_PAIR", "1")
    assert ops.pair_chain_ok(x, [w_ for w_, _ in layers], [b for _, b in layers], tuple(ops.L.ACT_IDS[a] for a in acts), "fp32")
    got = _run(ops, x, layers, acts, go)
    l64 = [(w_.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)) for w_, b in layers]
    x64 = x.detach().double().requires_grad_(True)
    h = x64
    for i, (w_, b) in enumerate(l64):
        h = h @ w_.T + b
        h = torch.relu(h) if i < nh else (torch.sigmoid(h) if out_act == "sigmoid" else h)
    (h * go.double()).sum().backward()
    g64 = [t_.grad for wb in l64 for t_ in wb]
    names = ["dx"] + [f"{'wb'[j]}{i}" for i in range(nh + 1) for j in range(2)]
    worst = ("", 0.0)
    for name, a, c in zip(names, [got[1]] + got[2], [x64.grad] + g64):
        assert bool(torch.isfinite(a).all()), name
        e = float((a.double() - c).abs().max()) / float(c.abs().max())
        worst = max(worst, (name, e), key=lambda p: p[1])
        assert e < 1e-5, (name, e)
    # dx row by row: a row whose cotangent is 1e-6 of the largest still gets a gradient with relative accuracy (the image
    # keeps 22 bits down to 2^-15 of the bound and 11 bits to 2^-28)
    rows = (w > 0) & (w > 1e-6 * w.max())
    dx, dx64 = got[1].double().cpu()[rows], x64.grad.cpu()[rows]
    rel = (dx - dx64).abs().amax(1) / dx64.abs().amax(1).clamp_min(1e-300)
    print(f"K {K} nh {nh} N2 {N2} dense {dense}: n {n}, weight spread {spread:.1e}, worst tensor {worst[0]} {worst[1]:.1e}, "
          f"worst dx row (weights > 1e-6 of the largest) {float(rel.max()):.1e}")
    assert float(rel.max()) < 2e-2
