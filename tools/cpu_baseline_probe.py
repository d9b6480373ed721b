"""Which host thread counts make the CPU oracle fastest on the GPU node (bench.py's cpu_baseline picks from this)."""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
import bench  # noqa: E402

omp = ctypes.CDLL("libgomp.so.1")


def main():
    import rise_sdf_amd as R
    torch.manual_seed(0)
    model = R.make("neus", bench.c1_config())
    model.geometry.update_step(0, 0)
    from helpers import camera_rays
    from test_gpu_model import oracle_params
    rays = camera_rays(800, 800, seed=0)[320000:320000 + 256].contiguous()
    u = torch.rand(256, generator=torch.Generator().manual_seed(2))
    meta, table, mlp, var = oracle_params(model)
    roi = torch.tensor([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5])
    ri, ts, te = oracle.ray_marching(rays[:, :3].contiguous(), rays[:, 3:].contiguous(), scene_aabb=roi,
                                     near_plane=0.0, far_plane=1e10, render_step_size=model.render_step_size,
                                     stratified_u=u)
    print("samples", ri.numel(), "cores", os.cpu_count())
    for tt, ot in [(8, 8), (16, 16), (32, 32), (64, 64), (32, 128), (64, 256), (128, 128), (256, 256)]:
        torch.set_num_threads(tt)
        omp.omp_set_num_threads(ot)
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            ref = oracle.neus_geometry_render(rays, ri, ts, te, table, meta, mlp, var, radius=1.5,
                                              fd_eps=model.geometry._finite_difference_eps)
            t1 = time.perf_counter()
            (ref["opacity"].sum() + ref["depth"].sum() + ref["comp_normal"].sum()).backward()
            t2 = time.perf_counter()
            best = min(best, t2 - t0)
        print(f"torch {tt:4d} omp {ot:4d}: fwd {t1 - t0:.2f} s  bwd {t2 - t1:.2f} s  best total {best:.2f} s  "
              f"{ri.numel() / best:.3g} samples/s", flush=True)


if __name__ == "__main__":
    main()
