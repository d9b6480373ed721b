#!/usr/bin/env python3
"""Algorithmic error of the split-operand matrix products on the CPU (numpy, products and sums in fp64: what the SCHEME
loses, without the fp32 accumulation rounding every scheme shares), for one layer z = W x of the SDF network's shapes:
    fp32 GEMM                          the reference (its own accumulation rounding, for scale)
    three bf16 parts, six products     rounds 1-3 (csrc/split_bf16.h)
    two fp16 parts, three products     round 4 (csrc/mlp_x2.hip), class scales 2^8 (inputs) / 2^6 (weights); with the fp16
                                       subnormals flushed to zero as well (the matrix cores keep them: tools/mfma_f16_subnormal.hip)
    python tools/accuracy_split.py"""
import numpy as np
import torch

rng = np.random.default_rng(0)


def bf16(x):
    return torch.from_numpy(x.astype(np.float32)).to(torch.bfloat16).to(torch.float32).numpy().astype(np.float64)


def split_bf16(x):
    x = x.astype(np.float32).astype(np.float64)
    h = bf16(x)
    m = bf16(x - h)
    return h, m, bf16(x - h - m)


def f16(x):
    return x.astype(np.float32).astype(np.float16).astype(np.float64)


def split_f16(x, scale):
    x = x.astype(np.float32).astype(np.float64) * scale
    h = f16(x)
    return h, f16(x - h)


def ftz(a):
    a = a.copy()
    a[np.abs(a) < 2.0 ** -14] = 0
    return a


K, N, H = 36, 8192, 64
for name, xs in [("features ~1e-4 (initialisation)", 1e-4), ("features ~0.05", 0.05), ("features ~3", 3.0)]:
    X = np.concatenate([rng.uniform(-1, 1, (32, N)) * xs * rng.uniform(0.01, 1, (32, N)), rng.uniform(-1, 1, (3, N)),
                        np.ones((1, N))]).astype(np.float32).astype(np.float64)
    W = rng.normal(0, 0.4, (H, K)).astype(np.float32).astype(np.float64)
    exact = W @ X
    sc = np.abs(exact).max()
    f32 = (W.astype(np.float32) @ X.astype(np.float32)).astype(np.float64)
    ah, am, al = split_bf16(W)
    bh, bm, bl = split_bf16(X)
    z6 = al @ bh + ah @ bl + am @ bm + am @ bh + ah @ bm + ah @ bh
    sw, sx = 2.0 ** 6, 2.0 ** 8
    wh, wl = split_f16(W, sw)
    xh, xl = split_f16(X, sx)
    z3 = (wl @ xh + wh @ xl + wh @ xh) / (sw * sx)
    z3f = (ftz(wl) @ ftz(xh) + ftz(wh) @ ftz(xl) + ftz(wh) @ ftz(xh)) / (sw * sx)
    e = lambda z: np.abs(z - exact).max() / sc
    print(f"{name:34s} max |error| / max |z|:  fp32 GEMM {e(f32):.2e}   bf16 x3, 6 products {e(z6):.2e}   "
          f"fp16 x2, 3 products {e(z3):.2e}   (subnormals flushed: {e(z3f):.2e})")
