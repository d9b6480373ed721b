#!/usr/bin/env python3
"""Debug aid: which operand of which radiance network leaves the pair kernels' range in the synthetic training step?
    python tools/debug/c3_range.py [steps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from rise_sdf_amd import ops, _lib as L           # noqa: E402
from rise_sdf_amd.step import build_synthetic_training   # noqa: E402

dev = torch.device("cuda:0")
model, ts = build_synthetic_training(dev, stage=1, hidden=128)
orig = ops.mlp_chain
seen = {}


def spy(x, layers, acts, *a, **kw):
    x2 = kw.get("x2")
    ws = [w for w, _ in layers]
    rec = {"n": x.shape[0], "x_max": float(x.abs().max()) if x.numel() else 0.0, "x_finite": bool(torch.isfinite(x).all()),
           "x2_max": None if x2 is None else float(x2.abs().max()), "x2_finite": None if x2 is None else bool(torch.isfinite(x2).all()),
           "w_max": max(float(w.abs().max()) for w in ws)}
    bad = (not rec["x_finite"]) or rec["x_max"] >= 1023 or rec["w_max"] >= 1023 or (x2 is not None and (not rec["x2_finite"] or rec["x2_max"] >= 1023))
    if bad:
        if not rec["x_finite"]:
            rows = (~torch.isfinite(x)).any(1).nonzero().flatten()
            rec["bad_rows"] = (int(rows.numel()), rows[:5].tolist(), rows[-5:].tolist())
        print("OUT OF RANGE", tuple(x.shape), None if x2 is None else tuple(x2.shape), rec, flush=True)
    return orig(x, layers, acts, *a, **kw)


ops.mlp_chain = spy
import rise_sdf_amd.network_utils as NU   # noqa: E402
if hasattr(NU, "mlp_chain"):
    NU.mlp_chain = spy
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    try:
        r = ts.step(20000 + k)
        print(k, r["num_rays"], r["num_samples"], flush=True)
        for name, prm in model.named_parameters():
            if not bool(torch.isfinite(prm).all()):
                print("   non-finite parameter", name, tuple(prm.shape), int((~torch.isfinite(prm)).sum()), flush=True)
            elif prm.grad is not None and not bool(torch.isfinite(prm.grad).all()):
                print("   non-finite grad", name, tuple(prm.shape), int((~torch.isfinite(prm.grad)).sum()), flush=True)
    except L.RiseSdfHipError as e:
        print(k, "ERROR", str(e)[:330], flush=True)
        L.poll_status(raise_on_error=False)
