#!/usr/bin/env python3
"""One c1 step at a given width / chunk size with the device's free memory sampled from a second thread: the lowest free
figure next to what the caching allocator holds tells how much HBM is taken OUTSIDE the allocator (kernel scratch, code
objects).    python tools/mem_probe.py [--hidden 128] [--chunk 24576] [--rays 49152]"""
import argparse
import json
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--chunk", type=int, default=24576)
    ap.add_argument("--rays", type=int, default=49152)
    ap.add_argument("--precision", default="fp32")
    args = ap.parse_args()
    import bench
    from rise_sdf_amd.ray_utils import orbit_view_rays
    dev = torch.device("cuda:0")
    model = bench.build_model(dev, argparse.Namespace(hidden=args.hidden, precision=args.precision))
    rays = orbit_view_rays(800, 800, seed=0, device=dev)
    n0 = (rays.shape[0] // 2 // 800) * 800
    rays = rays[n0:n0 + args.rays].contiguous()
    g = torch.Generator().manual_seed(2)
    jitter = torch.rand(rays.shape[0], generator=g).to(dev)
    cot = [torch.randn(rays.shape[0], 1, generator=g).to(dev), torch.randn(rays.shape[0], 1, generator=g).to(dev),
           torch.randn(rays.shape[0], 3, generator=g).to(dev)]
    lo = {"free": 1 << 60, "reserved_then": 0, "stop": False}

    def poll():
        while not lo["stop"]:
            f = torch.cuda.mem_get_info(dev)[0]
            if f < lo["free"]:
                lo["free"], lo["reserved_then"] = f, torch.cuda.memory_reserved(dev)
            time.sleep(0.002)

    th = threading.Thread(target=poll, daemon=True)
    th.start()
    out = {}
    try:
        res = bench.measure_c1(model, rays, jitter, cot, args.chunk, 1, 0, streams=1)
        out["samples_per_s"] = res["samples"] / res["dt"]
    except Exception as e:   # noqa: BLE001
        out["error"] = f"{type(e).__name__}: {e}"[:200]
    lo["stop"] = True
    th.join()
    total = torch.cuda.mem_get_info(dev)[1]
    out.update({"min_free_gib": round(lo["free"] / 2 ** 30, 1), "torch_reserved_then_gib": round(lo["reserved_then"] / 2 ** 30, 1),
                "outside_allocator_gib": round((total - lo["free"] - lo["reserved_then"]) / 2 ** 30, 1),
                "torch_peak_reserved_gib": round(torch.cuda.max_memory_reserved(dev) / 2 ** 30, 1)})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
