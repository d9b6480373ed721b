#!/usr/bin/env python3
"""Times the per-layer MLP entry points (rsdf_linear_fwd / _bwd_input / _bwd_weight / _bwd_fused) alone on the GPU, for the
layer shapes of the drop-in route's SDF network (35 -> 64 -> 64 -> 49 on [7 S] rows) and of the radiance networks
(84 -> 128 -> 128 -> 6): ms per launch, rows/s and the HBM bytes per second the call's operands amount to.
    python tools/bench_linear.py [--rows 8000000] [--reps 5]            (A/B builds: RSDF_LIB, tools/ab_mlp.sh)"""
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
    ap.add_argument("--rows", type=int, default=8_000_000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--shapes", default="35x64,64x64,64x49,84x128,128x128,128x6")
    args = ap.parse_args()
    from rise_sdf_amd import _lib
    L, P, st = _lib.lib(), _lib.ptr, _lib.stream_ptr()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    n = args.rows
    relu = _lib.ACT_IDS["relu"]
    out = {}

    def timed(fn):
        assert fn() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.reps

    for shp in args.shapes.split(","):
        K, N = (int(v) for v in shp.split("x"))
        x = torch.randn(n, K, generator=g).to(dev)
        w = (torch.randn(N, K, generator=g) * 0.1).to(dev)
        b = torch.zeros(N, device=dev)
        y = torch.empty(n, N, device=dev)
        gy = torch.randn(n, N, generator=g).to(dev)
        dz = torch.empty(n, N, device=dev)
        dx = torch.empty(n, K, device=dev)
        dw = torch.zeros(N, K, device=dev)
        db = torch.zeros(N, device=dev)
        r = {}
        t = timed(lambda: L.rsdf_linear_fwd(P(x), K, P(w), P(b), n, K, N, relu, P(y), N, st))
        r["fwd_ms"], r["fwd_TBps"] = round(t, 3), round(4e-9 * n * (K + N) / t, 2)
        t = timed(lambda: L.rsdf_linear_bwd_input(P(gy), P(y), N, P(w), n, K, N, relu, 0, K, P(dz), P(dx), K, st))
        r["bwd_input_ms"], r["bwd_input_TBps"] = round(t, 3), round(4e-9 * n * (3 * N + K) / t, 2)
        t = timed(lambda: L.rsdf_linear_bwd_weight(P(dz), N, P(x), K, n, K, N, P(dw), P(db), st))
        r["bwd_weight_ms"], r["bwd_weight_TBps"] = round(t, 3), round(4e-9 * n * (N + K) / t, 2)
        if L.rsdf_linear_bwd_fused_supported(K, N):
            t = timed(lambda: L.rsdf_linear_bwd_fused(P(gy), P(y), N, P(x), K, P(w), n, K, N, relu, 0, K, P(dx), K,
                                                      _lib.ACT_IDS["none"], P(dw), P(db), st))
            r["bwd_fused_ms"], r["bwd_fused_TBps"] = round(t, 3), round(4e-9 * n * (2 * N + 2 * K) / t, 2)
        out[shp] = r
        del x, y, gy, dz, dx
    print(json.dumps({"rows": n, **out}))


if __name__ == "__main__":
    main()
