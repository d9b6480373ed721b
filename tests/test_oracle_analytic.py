"""CPU: the twice-differentiable torch statement of the hash grid (oracle/analytic.py) against the C oracle,
and finite-difference checks of the analytic gradient / curvature restatements."""
import torch

import oracle
from oracle import analytic as A


def _setup(seed=0, n_levels=6):
    meta, n = oracle.grid_meta(n_levels=n_levels, base_resolution=16, log2_hashmap_size=12)
    g = torch.Generator().manual_seed(seed)
    table = (torch.rand(n, generator=g) * 2 - 1) * 0.1
    return meta, table, g


def test_torch_hashgrid_equals_c_oracle():
    meta, table, g = _setup()
    x = torch.rand(600, 3, generator=g)
    x[:4] = torch.tensor([[0.0, 0, 0], [1.0, 1, 1], [0.5, 0.5, 0.5], [1.0, 0.0, 0.25]])
    ref = oracle.hashgrid_encode(x, table, meta)
    got = A.hashgrid_encode_t(x.double(), table.double(), meta)
    assert float((ref.double() - got).abs().max()) < 2e-6
    got4 = A.hashgrid_encode_t(x.double(), table.double(), meta, n_active_levels=4)
    assert float(got4[:, 8:].abs().max()) == 0.0 and torch.equal(got4[:, :8], got[:, :8])


def test_analytic_gradient_matches_central_differences():
    meta, table, g = _setup(1)
    mlp = oracle.sphere_init_mlp_params(3 + 12, 13, 32, 2, seed=3)
    with torch.no_grad():
        mlp[0]["v"][:, 3:] = torch.randn(32, 12, generator=g) * 0.5
    mlp = [{k: v.double().requires_grad_(True) for k, v in p.items()} for p in mlp]
    pts = (torch.rand(200, 3, generator=g, dtype=torch.float64) * 2 - 1) * 1.4
    sdf, grad, feat = A.volume_sdf_analytic(pts, table.double(), meta, mlp, radius=1.5)
    h = 1e-6
    num = torch.stack([(A.field(pts + h * torch.eye(3, dtype=torch.float64)[d], table.double(), meta, mlp, 1.5)[:, 0]
                        - A.field(pts - h * torch.eye(3, dtype=torch.float64)[d], table.double(), meta, mlp, 1.5)[:, 0])
                       / (2 * h) for d in range(3)], -1)
    # trilinear interpolation is C0 across cell faces: exclude samples within h of a face at any level
    ok = (grad.detach() - num).abs().max(-1).values < 1e-4 * (1 + num.abs().max())
    assert int(ok.sum()) >= 195
    lap = A.curvature(pts, grad, torch.rand(200, 3, generator=g, dtype=torch.float64), table.double(), meta, mlp,
                      radius=1.5)
    assert lap.shape == (200,) and bool(((lap >= 0) & (lap <= 1)).all())
    (gt,) = torch.autograd.grad(lap.sum(), mlp[0]["v"], allow_unused=True)
    assert gt is not None and bool(torch.isfinite(gt).all())
