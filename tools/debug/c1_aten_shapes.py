"""Debug: the aten operators of one c1 chunk (bench.py's model, 28672 rays) by device time, with input shapes."""
import argparse, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from torch.profiler import profile, ProfilerActivity
from rise_sdf_amd.ray_utils import orbit_view_rays
dev = torch.device("cuda", 0)
model = bench.build_model(dev, argparse.Namespace(hidden=64))
rays = orbit_view_rays(800, 800, seed=0, device=dev)[320000:320000 + 2 * 28672].contiguous()
u = torch.rand(rays.shape[0], device=dev)
cot = [torch.randn(rays.shape[0], 1, device=dev), torch.randn(rays.shape[0], 1, device=dev), torch.randn(rays.shape[0], 3, device=dev)]
bench.run_step(model, rays, u, cot, 28672, 1)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    bench.run_step(model, rays, u, cot, 28672, 1)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    dt = getattr(e, "self_device_time_total", None)
    if dt is None:
        dt = getattr(e, "self_cuda_time_total", 0)
    if dt > 0 and e.key.startswith("aten::"):
        rows.append((dt, e.count, e.key, str(e.input_shapes)[:120]))
rows.sort(reverse=True)
print("aten ops of two c1 chunks by self device time (us, calls, op, shapes); total %.0f us" % sum(r[0] for r in rows))
for r in rows[:25]:
    print("%10.1f %5d  %-28s %s" % r)
