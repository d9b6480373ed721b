import ctypes, sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import torch, numpy as np
import oracle
from rise_sdf_amd import _lib, ops
from test_gpu_ops import _stencil_points, GRIDS
dev = torch.device('cuda:0')
cfg = GRIDS[1]
meta_g, n_params = _lib.make_grid_meta(**cfg)
radius, S = 1.5, 5000
for eps_cells in (1.0, 6.35, 3.0):
    eps = 2 * radius / 8192 * eps_cells
    tg = ((torch.rand(n_params, generator=torch.Generator().manual_seed(5)) * 2 - 1) * 1e-4).to(dev)
    x7t, pts = _stencil_points(dev, ops, S, eps, radius)
    a = torch.empty(16, 7, S, 2, device=dev); b = torch.full_like(a, float('nan'))
    _lib.lib().rsdf_hashgrid_fwd_fd7(_lib.ptr(x7t), _lib.ptr(tg), ctypes.byref(meta_g), S, 16, _lib.ptr(a), _lib.stream_ptr())
    _lib.lib().rsdf_hashgrid_fwd_fd7_pts(_lib.ptr(pts), radius, eps, _lib.ptr(tg), ctypes.byref(meta_g), S, 16, _lib.ptr(b), _lib.stream_ptr())
    torch.cuda.synchronize()
    bad = (a != b).any(-1)
    idx = bad.nonzero()
    print(eps_cells, 'mismatches', idx.shape[0], 'of', bad.numel())
    print(' per level', bad.sum((1, 2)).tolist()); print(' per tap', bad.sum((0, 2)).tolist())
    print(' samples', sorted(set(idx[:, 2].tolist()))[:20])
    for l, t, s in idx[:5].tolist():
        print('  l,t,s', l, t, s, 'p', pts[s].tolist(), 'x7t', x7t[t, s].tolist(), a[l, t, s].tolist(), b[l, t, s].tolist())
