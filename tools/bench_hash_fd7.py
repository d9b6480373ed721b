#!/usr/bin/env python3
"""Times the stencil hash kernels of one c1 chunk (32768 rays, ~18.9 M samples, L=16 T=2^19) in both stencil-source
forms: x7t (tap positions read once per level) and pts (taps derived in-kernel from the world-space centre).
    python tools/bench_hash_fd7.py [--rays 32768] [--reps 5] [--only fwd|bwd]
Prints one JSON line; used for A/B builds (tools/ab_hash2.sh) and the PMC passes (tools/pmc_passes.sh)."""
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
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--forms", default="x7t,pts,x2")
    ap.add_argument("--shell", type=float, default=0.0,
                    help="keep only the samples within this distance of the sphere |p| = 0.5 and shuffle the RAYS first (the "
                         "training step's sample set: random pixels, visibility-pruned to a thin shell)")
    args = ap.parse_args()
    import bench
    from rise_sdf_amd import _lib, ops
    from rise_sdf_amd.ray_utils import orbit_view_rays
    dev = torch.device("cuda:0")
    model = bench.build_model(dev, argparse.Namespace(hidden=64))
    rays = orbit_view_rays(800, 800, seed=0, device=dev)
    n0 = (rays.shape[0] // 2 // 800) * 800
    if args.shell > 0:
        rays = rays[torch.randperm(rays.shape[0], generator=torch.Generator().manual_seed(1)).to(dev)]
    rays = rays[n0:n0 + args.rays].contiguous()
    u = torch.rand(rays.shape[0], generator=torch.Generator().manual_seed(2)).to(dev)
    geo = model.geometry
    with torch.no_grad():
        ro, rd = rays[:, :3].contiguous(), rays[:, 3:].contiguous()
        ri, ts, te = model.occupancy_grid.sampling(ro, rd, render_step_size=model.render_step_size, stratified_u=u,
                                                   cone_angle=0.0, alpha_thre=0.0)
        x7t, pts = ops.fd_points(ro, rd, ri, ts, te, geo.radius, geo._finite_difference_eps, want_positions=True,
                                 tap_major=True)
    if args.shell > 0:
        keep = ((pts.norm(dim=-1) - 0.5).abs() < args.shell).nonzero().view(-1)
        pts, x7t = pts[keep].contiguous(), x7t[:, keep].contiguous()
    S = pts.shape[0]
    grid, _ = geo.encoding._hash()
    meta, table = grid.meta, grid.params.detach()
    Lv = int(meta.n_levels)
    radius, eps = float(geo.radius), float(geo._finite_difference_eps)
    eps_unit = eps / (2 * radius)
    planes = torch.empty(Lv, 7, S, 2, device=dev)
    dpl = torch.randn(Lv, 7, S, 2, device=dev)
    dt = torch.zeros(table.numel(), device=dev)
    nbytes = int(_lib.lib().rsdf_hashgrid_bwd_fd7_scratch_bytes(ctypes.byref(meta), S, Lv, eps_unit))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    st = _lib.stream_ptr()
    P = _lib.ptr
    L = _lib.lib()
    x2 = torch.empty(int(L.rsdf_x2_bytes(S, 2)), dtype=torch.uint8, device=dev)
    calls = {
        "fwd_x2": lambda: L.rsdf_hashgrid_fwd_fd7_x2(None, P(pts), radius, eps, P(table), ctypes.byref(meta), S, Lv, 2.0, -1.0,
                                                     2, P(x2), st),
        "fwd_x7t": lambda: L.rsdf_hashgrid_fwd_fd7(P(x7t), P(table), ctypes.byref(meta), S, Lv, P(planes), st),
        "fwd_pts": lambda: L.rsdf_hashgrid_fwd_fd7_pts(P(pts), radius, eps, P(table), ctypes.byref(meta), S, Lv,
                                                       P(planes), st),
        "bwd_x7t": lambda: L.rsdf_hashgrid_bwd_fd7(P(x7t), P(dpl), ctypes.byref(meta), S, Lv, eps_unit, P(dt),
                                                   P(scratch), nbytes, st),
        "bwd_pts": lambda: L.rsdf_hashgrid_bwd_fd7_pts(P(pts), radius, eps, P(dpl), ctypes.byref(meta), S, Lv,
                                                       eps_unit, P(dt), P(scratch), nbytes, st),
    }
    out = {"samples": S}
    for name, fn in calls.items():
        kind, form = name.split("_")
        if (args.only and kind != args.only) or form not in args.forms.split(","):
            continue
        assert fn() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[name + "_ms"] = round(e0.elapsed_time(e1) / args.reps, 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
