#!/usr/bin/env python3
"""Do the stencil gather (L2-request bound) and the fused MLP forward (vector-issue bound) overlap when they run on two
streams?  Times each alone and both together on one c1 chunk.   python tools/overlap_probe.py [--rays 32768]"""
import argparse
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=32768)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--hidden", type=int, default=64)
    args = ap.parse_args()
    import bench
    from rise_sdf_amd import _lib, ops
    from rise_sdf_amd.ray_utils import orbit_view_rays
    dev = torch.device("cuda:0")
    model = bench.build_model(dev, argparse.Namespace(hidden=args.hidden))
    rays = orbit_view_rays(800, 800, seed=0, device=dev)
    n0 = (rays.shape[0] // 2 // 800) * 800
    rays = rays[n0:n0 + args.rays].contiguous()
    u = torch.rand(rays.shape[0], generator=torch.Generator().manual_seed(2)).to(dev)
    geo = model.geometry
    with torch.no_grad():
        ro, rd = rays[:, :3].contiguous(), rays[:, 3:].contiguous()
        ri, ts, te = model.occupancy_grid.sampling(ro, rd, render_step_size=model.render_step_size, stratified_u=u,
                                                   cone_angle=0.0, alpha_thre=0.0)
        x7t, pts = ops.fd_points(ro, rd, ri, ts, te, geo.radius, geo._finite_difference_eps, want_positions=True,
                                 tap_major=True)
        ws = [t.detach().float().contiguous() for wb in geo.network.effective_weights() for t in wb]
    S = pts.shape[0]
    grid, _ = geo.encoding._hash()
    meta, table = grid.meta, grid.params.detach()
    Lv = int(meta.n_levels)
    radius, eps = float(geo.radius), float(geo._finite_difference_eps)
    H, N2 = ws[0].shape[0], ws[4].shape[0]
    planes_a = torch.empty(Lv, 7, S, 2, device=dev)
    planes_b = torch.randn(Lv, 7, S, 2, device=dev) * 1e-4
    sdf7t = torch.empty(7, S, device=dev)
    P, L = _lib.ptr, _lib.lib()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def G(st):
        return L.rsdf_hashgrid_fwd_fd7_pts(P(pts), radius, eps, P(table), ctypes.byref(meta), S, Lv, P(planes_a),
                                           ctypes.c_void_p(st.cuda_stream))

    def M(st):
        return L.rsdf_sdfmlp_fd7_fwd(P(x7t), P(planes_b), Lv, Lv, float(geo.encoding.xyz_scale),
                                     float(geo.encoding.xyz_offset), H, N2, *[P(t) for t in ws], S, P(sdf7t), None, None,
                                     ctypes.c_void_p(st.cuda_stream))

    def timed(fn):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        sa.wait_event(e0)
        sb.wait_event(e0)
        for _ in range(args.reps):
            fn()
        ea, eb = torch.cuda.Event(), torch.cuda.Event()
        ea.record(sa)
        eb.record(sb)
        torch.cuda.current_stream().wait_event(ea)
        torch.cuda.current_stream().wait_event(eb)
        e1.record()
        torch.cuda.synchronize()
        return round(e0.elapsed_time(e1) / args.reps, 3)

    assert G(sa) == 0 and M(sb) == 0
    out = {"samples": S, "H": H}
    out["gather_ms"] = timed(lambda: G(sa))
    out["mlp_fwd_ms"] = timed(lambda: M(sb))
    out["both_ms"] = timed(lambda: (M(sb), G(sa)))
    out["both_gather_first_ms"] = timed(lambda: (G(sa), M(sb)))
    out["sum_ms"] = round(out["gather_ms"] + out["mlp_fwd_ms"], 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
