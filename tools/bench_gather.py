"""The generic tcnn-shaped gather (what bench.py's "rsdf_hashgrid_fwd (generic)" entry times) in its two forms on the same
points: RSDF_GATHER=rows (one sample-group-major kernel writing rows) against RSDF_GATHER=staged (level-major planes, then
rows through LDS), on the centre points of one bench chunk and on uniformly random points, at several batch sizes.
    python tools/bench_gather.py [--chunk 24576]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def time_ms(fn, reps=5):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunk", type=int, default=24576)
    args = ap.parse_args()
    import bench
    from rise_sdf_amd import ops
    from rise_sdf_amd.ray_utils import orbit_view_rays
    dev = torch.device("cuda", 0)
    ns = argparse.Namespace(hidden=64, precision="fp32")
    model = bench.build_model(dev, ns)
    rays = orbit_view_rays(800, 800, seed=0, device=dev)
    jitter = torch.rand(rays.shape[0], generator=torch.Generator().manual_seed(2)).to(dev)
    with torch.no_grad():
        ro, rd = rays[:args.chunk, :3].contiguous(), rays[:args.chunk, 3:].contiguous()
        ri, ts, te = model.occupancy_grid.sampling(ro, rd, render_step_size=model.render_step_size,
                                                   stratified_u=jitter[:args.chunk], cone_angle=0.0, alpha_thre=0.0)
        x7 = ops.fd_points(ro, rd, ri, ts, te, model.geometry.radius, model.geometry._finite_difference_eps)
        centres = x7.view(-1, 7, 3)[:, 0].contiguous()
        del x7
        enc = model.geometry.encoding.encoding.encoding
        rnd = torch.rand(centres.shape[0], 3, generator=torch.Generator().manual_seed(5)).to(dev)
        for label, pts in (("ray-ordered centres", centres), ("uniform random", rnd)):
            for n in (1 << 16, 1 << 18, 1 << 20, 1 << 22, pts.shape[0]):
                x = pts[:n].contiguous()
                row = {}
                for mode in ("rows", "staged"):
                    os.environ["RSDF_GATHER"] = mode
                    row[mode] = time_ms(lambda: enc(x))
                os.environ.pop("RSDF_GATHER")
                print(f"{label:20s} n={x.shape[0]:9d}  rows {row['rows']:7.3f} ms ({x.shape[0] / row['rows'] / 1e6:6.2f} e9/s)"
                      f"   staged {row['staged']:7.3f} ms ({x.shape[0] / row['staged'] / 1e6:6.2f} e9/s)", flush=True)


if __name__ == "__main__":
    main()
