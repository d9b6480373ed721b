#!/usr/bin/env python3
"""The environment prefilter's kernels alone (rsdf_specular_cubemap_fwd_norm / _bwd, csrc/envlight.hip), per level of the
yaml's 512^2 light (lib/pbr/light.py:177-180), timed with events on an otherwise idle GPU:
    python tools/bench_prefilter.py [--reps 20]
One line: fwd / bwd ms per level and their sums (what a training step pays per build_mips + its backward)."""
import argparse, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    from rise_sdf_amd import envlight as E
    from rise_sdf_amd._lib import check, lib, ptr, stream_ptr
    dev = torch.device("cuda", 0)
    levels = [(512, 0.08), (256, 0.185), (128, 0.29), (64, 0.395), (32, 0.5)]
    out = {}
    for R, rough in levels:
        c = torch.rand(6, R, R, 3, device=dev)
        g = torch.randn(6, R, R, 3, device=dev)
        cosc, bounds = E.specular_bounds(R, rough, 0.99, dev)
        table = E.texel_table(R, dev)
        o3, ws, gc = torch.empty(6, R, R, 3, device=dev), torch.empty(6, R, R, 1, device=dev), torch.empty(6, R, R, 3, device=dev)
        def fwd():
            check(lib().rsdf_specular_cubemap_fwd_norm(ptr(c), ptr(bounds), ptr(table), R, float(rough), float(cosc), ptr(o3), ptr(ws), stream_ptr()), "f")
        def bwd():
            check(lib().rsdf_specular_cubemap_bwd(ptr(g), 3, ptr(bounds), ptr(table), R, float(rough), float(cosc), ptr(gc), stream_ptr()), "b")
        res = []
        for fn in (fwd, bwd):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(args.reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res.append(round(e0.elapsed_time(e1) / args.reps, 4))
        out[f"{R}"] = res
    out["fwd_ms"] = round(sum(v[0] for v in out.values() if isinstance(v, list)), 3)
    out["bwd_ms"] = round(sum(v[1] for k, v in out.items() if isinstance(v, list)), 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
