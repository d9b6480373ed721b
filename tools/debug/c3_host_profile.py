#!/usr/bin/env python3
"""Debug aid: where does the HOST spend the training step?  cProfile over 20 steps after the settle phase (relative shares only:
the profiler itself slows Python frames down), plus the wall time of the same steps with and without a device drain per step.
    python tools/debug/c3_host_profile.py"""
import cProfile
import os
import pstats
import sys
import time

import torch

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
t0 = time.perf_counter()
for k in range(20):
    ts.step(gs + k)
t_host = time.perf_counter() - t0          # host time to ENQUEUE 20 steps (the one read per step drains part of it)
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"20 steps: host returns after {t_host * 50:.2f} ms/step, device done after {t_all * 50:.2f} ms/step")
gs += 20
pr = cProfile.Profile()
pr.enable()
for k in range(20):
    ts.step(gs + k)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
