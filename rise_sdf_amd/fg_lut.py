"""Pre-integrated GGX split-sum table ("FG LUT").

The reference reads ``load/bsdf/bsdf_256_256.bin`` (fp32 [256,256,2], models/texture.py:285-287), a file that
is not part of its repository (README.md:68, .MISSING_LARGE_BLOBS).  When the file exists it is used as is;
otherwise the table is integrated here: for NoV = (i+1/2)/256 (x axis) and roughness = (j+1/2)/256 (y axis),
    A = int f(l) (1 - Fc) G_vis NoL,  B = int Fc G_vis NoL      (Karis 2013, importance-sampled GGX),
with alpha = roughness^2, Fc = (1 - VoH)^5 and the Smith-GGX correlated visibility term.  Host-side, one-off
(numpy, deterministic Hammersley points); the lookup itself is the HIP grid_sample kernel.
"""
from __future__ import annotations

import os

import numpy as np
import torch

_cache = {}


def integrate_fg_lut(res=256, n_samples=256):
    i = (np.arange(n_samples) + 0.5) / n_samples
    # radical inverse base 2 (Hammersley second coordinate)
    bits = np.arange(n_samples, dtype=np.uint32)
    rev = np.zeros(n_samples, dtype=np.float64)
    f = 0.5
    b = bits.copy()
    while b.any():
        rev += f * (b & 1)
        b >>= 1
        f *= 0.5
    u1, u2 = i[None, None, :], rev[None, None, :]
    nov = ((np.arange(res) + 0.5) / res)[None, :, None]          # x axis
    rough = ((np.arange(res) + 0.5) / res)[:, None, None]        # y axis
    a = rough * rough
    a2 = a * a
    phi = 2 * np.pi * u1
    cos_t = np.sqrt((1 - u2) / (1 + (a2 - 1) * u2))
    sin_t = np.sqrt(np.maximum(0.0, 1 - cos_t * cos_t))
    hx, hz = sin_t * np.cos(phi), cos_t
    vx, vz = np.sqrt(np.maximum(0.0, 1 - nov * nov)), nov
    voh = vx * hx + vz * hz
    lz = 2 * voh * hz - vz
    nol, noh, voh_c = np.clip(lz, 0, 1), np.clip(hz, 0, 1), np.clip(voh, 0, 1)
    # Smith-GGX height-correlated visibility * 4 NoL NoV / ... in the importance-sampled estimator form
    gv = nol * np.sqrt(nov * nov * (1 - a2) + a2)
    gl = nov * np.sqrt(nol * nol * (1 - a2) + a2)
    vis = 0.5 / np.maximum(gv + gl, 1e-8)
    g_vis = vis * voh_c * nol * 4.0 / np.maximum(noh, 1e-8)
    fc = (1 - voh_c) ** 5
    ok = nol > 0
    A = np.where(ok, (1 - fc) * g_vis, 0.0).mean(-1)
    B = np.where(ok, fc * g_vis, 0.0).mean(-1)
    return np.stack([A, B], -1).astype(np.float32)               # [rough (y), NoV (x), 2]


def load_or_build_fg_lut(path="load/bsdf/bsdf_256_256.bin"):
    if path and os.path.exists(path):
        return torch.from_numpy(np.fromfile(path, dtype=np.float32).reshape(1, 256, 256, 2))
    if "lut" not in _cache:
        _cache["lut"] = torch.from_numpy(integrate_fg_lut())[None]
    return _cache["lut"].clone()
