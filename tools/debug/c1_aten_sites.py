#!/usr/bin/env python3
"""Debug aid: the aten kernels of a c1 chunk with their Python call sites (forward) / autograd node (backward).
    python tools/debug/c1_aten_sites.py [width]"""
import os
import sys
import types

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench   # noqa: E402
from rise_sdf_amd.ray_utils import orbit_view_rays   # noqa: E402

dev = torch.device("cuda:0")
w = int(sys.argv[1]) if len(sys.argv) > 1 else 320
model = bench.build_model(dev, types.SimpleNamespace(hidden=64, precision="fp32"))
rays = orbit_view_rays(w, w, seed=0, device=dev)
n = rays.shape[0]
g = torch.Generator().manual_seed(2)
jitter = torch.rand(n, generator=g).to(dev)
cot = [torch.randn(n, 1, generator=g).to(dev), torch.randn(n, 1, generator=g).to(dev), torch.randn(n, 3, generator=g).to(dev)]
bench.run_step(model, rays, jitter, cot, 28672)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    bench.run_step(model, rays, jitter, cot, 28672)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_stack_n=14, group_by_input_shape=True):
    self_t = getattr(e, "self_device_time_total", 0)
    if self_t <= 0 or not (e.key.startswith("aten::") or "Memcpy" in e.key or "Memset" in e.key):
        continue
    own = [s for s in e.stack if "/rise_sdf_amd/" in s or "bench.py" in s][:3]
    rows.append((self_t, e.count, e.key, str(e.input_shapes)[:70], own))
rows.sort(key=lambda r: -r[0])
for t, c, k, shp, own in rows[:25]:
    print(f"{t / 1e3:9.2f} ms x{c:<4d} {k:22s} {shp:70s} {' <- '.join(s.split('/')[-1][:60] for s in own)}")
