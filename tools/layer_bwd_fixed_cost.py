#!/usr/bin/env python3
"""Launch-size sweep of the one-pass 128-wide layer backward (rsdf_linear_bwd_fused) and of the per-layer forward: time per
launch against the row count, to separate the per-launch fixed cost (weight staging, the dW flush) from the per-row cost.
    python tools/layer_bwd_fixed_cost.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from rise_sdf_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    w = [(torch.randn(128, 128, generator=g) * 0.1).to(dev).requires_grad_(True) for _ in range(3)]
    b = [torch.zeros(128, device=dev, requires_grad=True) for _ in range(3)]
    out = {}
    for n in (64, 16384, 65536, 131072, 262144, 524288, 1048576, 4194304):
        x = torch.randn(n, 128, generator=g).to(dev).requires_grad_(True)
        gy = torch.randn(n, 128, generator=g).to(dev)
        def run():
            y = ops.mlp_chain(x, list(zip(w, b)), ["relu", "relu", "relu"])
            return y
        for _ in range(3):
            run().backward(gy)
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        reps = 20
        tf = tb = 0.0
        for _ in range(reps):
            e[0].record()
            y = run()
            e[1].record()
            y.backward(gy)
            e[2].record()
            torch.cuda.synchronize()
            tf += e[0].elapsed_time(e[1])
            tb += e[1].elapsed_time(e[2])
        out[n] = {"fwd_ms_per_layer": round(tf / reps / 3, 4), "bwd_ms_per_layer": round(tb / reps / 3, 4)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
