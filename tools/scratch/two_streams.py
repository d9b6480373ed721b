import sys, os, time, argparse
sys.path.insert(0, os.getcwd())
import torch
import bench
from rise_sdf_amd.ray_utils import orbit_view_rays
dev = torch.device('cuda:0')
ap = argparse.ArgumentParser(); ap.add_argument('--chunk', type=int, default=16384); ap.add_argument('--streams', type=int, default=2)
args = ap.parse_args()
model = bench.build_model(dev, argparse.Namespace(hidden=64, precision='fp32'))
rays = orbit_view_rays(800, 800, seed=0, device=dev)
n = rays.shape[0]
g = torch.Generator().manual_seed(2)
jitter = torch.rand(n, generator=g).to(dev)
cot = [torch.randn(n, 1, generator=g).to(dev), torch.randn(n, 1, generator=g).to(dev), torch.randn(n, 3, generator=g).to(dev)]
streams = [torch.cuda.Stream() for _ in range(args.streams)]

def step():
    for p in model.parameters(): p.grad = None
    total = 0
    torch.cuda.synchronize()
    for k, s in enumerate(range(0, n, args.chunk)):
        e = min(s + args.chunk, n)
        st = streams[k % len(streams)]
        with torch.cuda.stream(st):
            out = model.forward_(rays[s:e], stratified_u=jitter[s:e])
            total += int(out['ray_indices'].numel())
            torch.autograd.backward([out['opacity'], out['depth'], out['comp_normal_raw']], [cot[0][s:e], cot[1][s:e], cot[2][s:e]])
    torch.cuda.synchronize()
    return total
step()
t0 = time.perf_counter(); S = step() + step(); dt = time.perf_counter() - t0
print('chunk', args.chunk, 'streams', args.streams, '%.4g samples/s' % (S / dt), 'ms/step %.1f' % (dt / 2 * 1e3))
