"""T3: the one-pass backward of a 128-wide VanillaMLP layer (rsdf_linear_bwd_fused, mlp_layer_bwd.hip) against an fp64
restatement of models/network_utils.py:109-157's autograd, and against the two-kernel form it replaces."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

ACTS = {"none": 0, "relu": 1, "softplus100": 2, "sigmoid": 3}


def _ref(x, w, b, gy, act):
    x64, w64, b64 = (t.double().detach().requires_grad_(True) for t in (x, w, b))
    z = x64 @ w64.t() + b64
    if act == "relu":
        y = torch.relu(z)
    elif act == "softplus100":
        y = torch.nn.functional.softplus(z, beta=100, threshold=20)
    elif act == "sigmoid":
        y = torch.sigmoid(z)
    else:
        y = z
    y.backward(gy.double())
    return y.detach(), x64.grad, w64.grad, b64.grad


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


@pytest.mark.parametrize("K,k0,kout", [(128, 0, 128), (84, 0, 84), (84, 3, 81), (73, 0, 73), (48, 16, 20), (3, 0, 3)])
@pytest.mark.parametrize("act", ["relu", "softplus100", "none", "sigmoid"])
@pytest.mark.parametrize("n", [1, 63, 1000, 4097])
def test_layer_bwd_fused_matches_fp64(K, k0, kout, act, n):
    from rise_sdf_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(K * 131 + n)
    x = torch.randn(n, K, generator=g).to(dev)
    w = (torch.randn(128, K, generator=g) / K ** 0.5).to(dev)
    b = (0.1 * torch.randn(128, generator=g)).to(dev)
    gy = torch.randn(n, 128, generator=g).to(dev)
    y_ref, dx_ref, dw_ref, db_ref = _ref(x, w, b, gy, act)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = ops.linear(xr, wr, br, act=act, dx_cols=None if (k0, kout) == (0, K) else (k0, kout))
    assert _rel(y, y_ref) < 2e-6
    y.backward(gy)
    assert _rel(wr.grad, dw_ref) < 3e-6, ("dw", _rel(wr.grad, dw_ref))
    assert _rel(br.grad, db_ref) < 3e-6, ("db", _rel(br.grad, db_ref))
    win = dx_ref.clone()
    win[:, :k0] = 0
    win[:, k0 + kout:] = 0
    assert _rel(xr.grad, win) < 3e-6, ("dx", _rel(xr.grad, win))
    assert torch.all(xr.grad[:, :k0] == 0) and torch.all(xr.grad[:, k0 + kout:] == 0)


def test_layer_bwd_fused_equals_split_form(monkeypatch):
    """Same inputs through the two-kernel form (RSDF_LAYER_BWD=split): both are fp32-faithful, so they agree to rounding."""
    from rise_sdf_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(7)
    n, K = 5000, 84
    x = torch.randn(n, K, generator=g).to(dev)
    w = (torch.randn(128, K, generator=g) / K ** 0.5).to(dev)
    b = (0.1 * torch.randn(128, generator=g)).to(dev)
    gy = torch.randn(n, 128, generator=g).to(dev)
    outs = []
    for mode in ("fused", "split"):
        monkeypatch.setenv("RSDF_LAYER_BWD", mode)
        xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        ops.linear(xr, wr, br, act="relu").backward(gy)
        outs.append((xr.grad, wr.grad, br.grad))
    for a, bb in zip(*outs):
        assert _rel(a, bb) < 2e-6


def test_layer_bwd_fused_without_input_grad():
    """First layer of a network whose input needs no gradient: the dx-free instantiation."""
    from rise_sdf_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(11)
    n, K = 777, 96
    x = torch.randn(n, K, generator=g).to(dev)
    w = (torch.randn(128, K, generator=g) / K ** 0.5).to(dev)
    b = torch.zeros(128).to(dev)
    gy = torch.randn(n, 128, generator=g).to(dev)
    _, _, dw_ref, db_ref = _ref(x, w, b, gy, "relu")
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ops.linear(x, wr, br, act="relu").backward(gy)
    assert _rel(wr.grad, dw_ref) < 3e-6 and _rel(br.grad, db_ref) < 3e-6


def test_layer_bwd_fused_rejects_other_widths():
    from rise_sdf_amd import _lib
    lib = _lib.lib()
    assert lib.rsdf_linear_bwd_fused_supported(84, 128) == 1
    assert lib.rsdf_linear_bwd_fused_supported(128, 64) == 0
    p = ctypes.c_void_p(16)
    assert lib.rsdf_linear_bwd_fused(p, p, 64, p, 64, p, 10, 64, 64, 0, 0, 0, None, 0, 0, p, None, None) != 0


def _chain_ref(x, layers, acts, gy):
    x64 = x.double().detach().requires_grad_(True)
    ps = [(w.double().detach().requires_grad_(True), b.double().detach().requires_grad_(True)) for w, b in layers]
    h = x64
    for (w, b), a in zip(ps, acts):
        z = h @ w.t() + b
        h = {"relu": torch.relu, "sigmoid": torch.sigmoid, "none": lambda t: t,
             "softplus100": lambda t: torch.nn.functional.softplus(t, beta=100, threshold=20)}[a](z)
    h.backward(gy.double())
    return h.detach(), x64.grad, [(w.grad, b.grad) for w, b in ps]


@pytest.mark.parametrize("hidden_act,out_act", [("relu", "sigmoid"), ("relu", "none"), ("softplus100", "none")])
@pytest.mark.parametrize("dims", [(84, 128, 128, 128, 128, 3), (73, 128, 128, 1), (128, 128, 6), (35, 64, 64, 13)])
@pytest.mark.parametrize("n", [257, 5000])
def test_mlp_chain_matches_fp64(hidden_act, out_act, dims, n, monkeypatch):
    """The whole VanillaMLP as one autograd node (ops.mlp_chain): 128-wide ReLU layers hand dz to the layer below from
    inside the kernel; other widths / activations take the two-kernel layers.  All gradients vs fp64 autograd."""
    from rise_sdf_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(n + len(dims))
    x = torch.randn(n, dims[0], generator=g).to(dev)
    layers = [((torch.randn(o, i, generator=g) * (2.0 / i) ** 0.5).to(dev), (0.1 * torch.randn(o, generator=g)).to(dev))
              for i, o in zip(dims[:-1], dims[1:])]
    acts = [hidden_act] * (len(layers) - 1) + [out_act]
    gy = torch.randn(n, dims[-1], generator=g).to(dev)
    y_ref, dx_ref, gref = _chain_ref(x, layers, acts, gy)
    for mode in ("fused", "split"):
        monkeypatch.setenv("RSDF_LAYER_BWD", mode)
        xr = x.clone().requires_grad_(True)
        ps = [(w.clone().requires_grad_(True), b.clone().requires_grad_(True)) for w, b in layers]
        y = ops.mlp_chain(xr, ps, acts)
        assert _rel(y, y_ref) < 5e-6
        y.backward(gy)
        assert _rel(xr.grad, dx_ref) < 1e-5, (mode, "dx", _rel(xr.grad, dx_ref))
        for (w, b), (gw, gb) in zip(ps, gref):
            assert _rel(w.grad, gw) < 1e-5, (mode, "dw", tuple(w.shape), _rel(w.grad, gw))
            assert _rel(b.grad, gb) < 1e-5, (mode, "db", tuple(w.shape), _rel(b.grad, gb))


def test_mlp_chain_input_window_and_frozen_input():
    from rise_sdf_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(3)
    n, dims = 1500, (35, 128, 128, 13)
    x = torch.randn(n, dims[0], generator=g).to(dev)
    layers = [((torch.randn(o, i, generator=g) * (2.0 / i) ** 0.5).to(dev), torch.zeros(o).to(dev))
              for i, o in zip(dims[:-1], dims[1:])]
    acts = ["relu", "relu", "none"]
    gy = torch.randn(n, dims[-1], generator=g).to(dev)
    _, dx_ref, gref = _chain_ref(x, layers, acts, gy)
    xr = x.clone().requires_grad_(True)
    ps = [(w.clone().requires_grad_(True), b.clone().requires_grad_(True)) for w, b in layers]
    ops.mlp_chain(xr, ps, acts, dx_cols=(3, 32)).backward(gy)
    assert torch.all(xr.grad[:, :3] == 0) and _rel(xr.grad[:, 3:], dx_ref[:, 3:]) < 1e-5
    ps2 = [(w.clone().requires_grad_(True), b.clone().requires_grad_(True)) for w, b in layers]
    ops.mlp_chain(x, ps2, acts).backward(gy)       # the input needs no gradient: the first layer runs the dx-free kernel
    for (w, b), (gw, gb) in zip(ps2, gref):
        assert _rel(w.grad, gw) < 1e-5 and _rel(b.grad, gb) < 1e-5


def test_weight_norm_cache_semantics():
    """VanillaMLP computes weight_norm(g, v) once per parameter version (network_utils._normed_weight): a second forward
    reuses W and its autograd node; a backward spends the node (the next forward re-normalises); an in-place parameter
    change is seen; a no_grad evaluation first does not hide the parameters from a later gradient pass."""
    import rise_sdf_amd as R
    from rise_sdf_amd.network_utils import get_mlp
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    cfg = R.Config({"otype": "VanillaMLP", "activation": "ReLU", "output_activation": "none", "n_neurons": 64,
                    "n_hidden_layers": 2, "sphere_init": True, "sphere_init_radius": 0.5, "weight_norm": True})
    net = get_mlp(35, 13, cfg).to(dev)
    x = torch.randn(300, 35, device=dev)
    with torch.no_grad():
        y0 = net(x)                                   # no_grad first: must not poison the cache for the gradient pass
    y1 = net(x)
    assert torch.equal(y0, y1) and y1.requires_grad
    y2 = net(2 * x)                                   # second evaluation in the same "step": same W node
    (y1.sum() + y2.sum()).backward()
    g_joint = [p.grad.clone() for p in net.parameters()]
    assert all(torch.isfinite(g).all() and g.abs().sum() > 0 for g in g_joint)
    net.zero_grad(set_to_none=True)
    y1 = net(x)                                       # after a backward without an optimizer step: the node was spent
    y1.sum().backward()
    y2 = net(2 * x)
    y2.sum().backward()                               # (would raise "backward through the graph a second time" if reused)
    for a, b in zip(g_joint, [p.grad for p in net.parameters()]):
        assert _rel(b, a) < 1e-5
    with torch.no_grad():                             # an in-place parameter change is seen by the next forward
        net.layers[0].weight_g.mul_(1.5)
    y3 = net(x)
    assert not torch.allclose(y3, y0)
