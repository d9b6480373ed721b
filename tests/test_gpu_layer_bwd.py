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
    assert lib.rsdf_linear_bwd_fused(p, p, 64, p, 64, p, 10, 64, 64, 0, 0, 0, None, 0, p, None, None) != 0
