#!/usr/bin/env python3
"""Debug aid: which Python lines own the aten kernels (fills, adds, sums, copies) of a config[2] step?
    python tools/debug/c2_aten_sites.py [width]"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_c2   # noqa: E402

dev = torch.device("cuda:0")
w = int(sys.argv[1]) if len(sys.argv) > 1 else 256
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    bench_c2.measure_c2(dev, width=w, height=w, steps=1)
rows = []
for e in prof.key_averages(group_by_stack_n=12):
    t = getattr(e, "device_time_total", None)
    if t is None:
        t = e.cuda_time_total
    self_t = getattr(e, "self_device_time_total", 0)
    if self_t <= 0:
        continue
    own = [s for s in e.stack if "/rise_sdf_amd/" in s or "/tools/" in s or "bench" in s][:3]
    rows.append((self_t, e.count, e.key, own))
rows.sort(key=lambda r: -r[0])
tot = sum(r[0] for r in rows)
print(f"total self device time {tot / 1e3:.1f} ms")
for t, c, k, own in rows[:40]:
    if k.startswith("aten::") or "Backward" in k or "hipMemcpy" in k or "Memcpy" in k or "Memset" in k:
        print(f"{t / 1e3:9.2f} ms  {100 * t / tot:5.1f}%  x{c:<5d} {k:32s} {' <- '.join(s.split('/')[-1][:70] for s in own)}")
