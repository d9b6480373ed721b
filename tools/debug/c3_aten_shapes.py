"""Debug: the aten operators of one training step (tools/bench_step.py's model at its operating point) by device time, with
input shapes (torch.profiler, record_shapes): which torch glue is worth folding into the HIP kernels."""
import os, sys, collections, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_step import build
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
model, ts = build(dev, stage=1, hidden=128)
for k in range(60):
    ts.step(20000 + k)
torch.cuda.synchronize()
steps = 5
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    for k in range(steps):
        ts.step(20060 + k)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    dt = getattr(e, "self_device_time_total", None)
    if dt is None:
        dt = getattr(e, "self_cuda_time_total", 0)
    if dt > 0 and e.key.startswith("aten::"):
        rows.append((dt / steps, e.count / steps, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
print("aten ops of one step by self device time (us per step, calls per step, op, input shapes); total %.0f us, %.0f calls" %
      (sum(r[0] for r in rows), sum(r[1] for r in rows)))
for r in rows[:int(os.environ.get("ATEN_ROWS", "70"))]:
    print("%9.1f %6.1f  %-34s %s" % r)
