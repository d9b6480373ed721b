#!/usr/bin/env python3
"""Debug aid: the aten kernels of the training step with input shapes and (forward ops) the Python call sites.
    python tools/debug/c3_aten_sites.py"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rise_sdf_amd.step import build_synthetic_training   # noqa: E402

dev = torch.device("cuda:0")
model, ts = build_synthetic_training(dev, stage=1, hidden=128)
gs = 20000
for k in range(80):
    ts.step(gs + k)
gs += 80
torch.cuda.synchronize()
N = 5
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for k in range(N):
        ts.step(gs + k)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_stack_n=16, group_by_input_shape=True):
    self_t = getattr(e, "self_device_time_total", 0)
    if self_t <= 0 or not (e.key.startswith("aten::") or "Memcpy" in e.key or "Memset" in e.key):
        continue
    own = [s for s in e.stack if "/rise_sdf_amd/" in s][:3]
    rows.append((self_t / N, e.count / N, e.key, str(e.input_shapes)[:60], own))
rows.sort(key=lambda r: -r[0])
print(f"aten device time per step: {sum(r[0] for r in rows) / 1e3:.2f} ms")
for t, c, k, shp, own in rows[:45]:
    print(f"{t:8.1f} us x{c:<5.1f} {k:20s} {shp:60s} {' <- '.join(s.split('/')[-1][:48] for s in own)}")
