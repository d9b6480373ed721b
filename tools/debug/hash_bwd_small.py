#!/usr/bin/env python3
"""Debug aid: the stencil hash backward on a training-sized batch (4096 rays, ~1e6 samples before pruning): coherent rays (a block of
one view) against rays drawn at random over the view, entry-point times by HIP events with nothing else on the GPU."""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench   # noqa: E402
from rise_sdf_amd import _lib   # noqa: E402
from rise_sdf_amd.ray_utils import orbit_view_rays   # noqa: E402

dev = torch.device("cuda:0")
model = bench.build_model(dev, types.SimpleNamespace(hidden=128, precision="fp32"))
rays = orbit_view_rays(800, 800, seed=0, device=dev)
g = torch.Generator().manual_seed(0)
sets = {"coherent 4096": rays[320000:324096], "random 4096": rays[torch.randperm(rays.shape[0], generator=g)[:4096].to(dev)],
        "coherent 1024": rays[320000:321024], "random 1024": rays[torch.randperm(rays.shape[0], generator=g)[:1024].to(dev)]}
for name, r in sets.items():
    n = r.shape[0]
    u = torch.rand(n, generator=g).to(dev)
    cot = [torch.randn(n, 1, generator=g).to(dev), torch.randn(n, 1, generator=g).to(dev), torch.randn(n, 3, generator=g).to(dev)]
    for rep in range(3):
        timer = _lib.KernelTimer()
        _lib.set_timer(timer)
        out = model.forward_(r.contiguous(), stratified_u=u)
        torch.autograd.backward([out["opacity"], out["depth"], out["comp_normal_raw"]], cot)
        torch.cuda.synchronize()
        _lib.set_timer(None)
    S = int(out["ray_indices"].numel())
    summ = timer.summary()
    top = {k: round(v["ms"], 3) for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:6]}
    print(f"{name}: {S} samples; {top}", flush=True)
