#!/usr/bin/env python3
"""bench.py -- ray-marched SDF samples/sec (fwd+bwd), BASELINE.json config[1].

    python bench.py --gpus N --steps K --warmup W

Workload ("c1"): one synthetic 800x800 pinhole view per rank (640,000 rays) of the
[-1.5,1.5]^3 box, dense marching (no occupancy pruning, 1024 samples per box diagonal), L=16 F=2
T=2^19 base-32 hash grid (14,533,536 fp32 params, U(-1e-4,1e-4)), 2x64 weight-normalised
Softplus(100) SDF MLP -> 48 features with the reference's sphere initialisation, finite-difference
normals (7 field evaluations per sample, eps = 3/8192), NeuS alpha, per-ray transmittance
compositing of opacity / depth / normals, and the full backward to every parameter.

A "step" is one pass over the rank's 640,000 rays in chunks of --chunk rays (forward + backward per
chunk, gradients accumulated), followed for N>1 by the RCCL mean-all-reduce of all gradients -- the
reference's DDP step (launch.py:84-97).  Weak scaling: every rank renders its own view.
A "sample" is one marched interval that is field-queried and composited (the reference's
``num_samples``, models/split_mixed_occ.py:349-351).

Rank 0 prints ONE JSON line; see DESIGN.md "Measurement" for the roofline definitions.
"""
import argparse
import json
import math
import os
import sys
import time

# The step allocates per-chunk buffers of 10-30 GB whose sizes follow the chunk's sample count (+-15 % from chunk to chunk).
# torch's caching allocator cannot serve a 13.0 GB request from a cached 12.9 GB block, so over the 23 chunks of a view its
# reserved memory crept up to 276 of the 288 GB (11 GiB free at the lowest point, one secondary variant out of memory);
# rounding request sizes up to quarter-power-of-two classes makes freed blocks reusable: 169 GiB reserved with two chunks in
# flight, 85 with one (``config.hbm_gib``).  Expandable segments are not available on this platform.  A value from the
# environment wins.
os.environ.setdefault("PYTORCH_HIP_ALLOC_CONF", "roundup_power2_divisions:4")
os.environ.setdefault("PYTORCH_CUDA_ALLOC_CONF", os.environ["PYTORCH_HIP_ALLOC_CONF"])

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

T_PROCESS_START = time.perf_counter()
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TF = 157.3   # MI355X_MICROARCH.md: fp32 matrix peak
MFMA_BF16_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: dense bf16 matrix peak (the sparsity-doubled headline is never used)


def c1_config(hidden=64, n_levels=16, log2_T=19, base=32, feat=48, precision="fp32"):
    from rise_sdf_amd import Config
    return Config({
        "name": "neus", "radius": 1.5, "num_samples_per_ray": 1024, "randomized": True,
        "ray_chunk": 4096, "cos_anneal_end": 0, "learned_background": False, "grid_prune": False,
        "variance": {"init_val": 0.3, "modulate": False},
        "geometry": {
            "name": "volume-sdf", "radius": 1.5, "feature_dim": feat,
            "grad_type": "finite_difference", "finite_difference_eps": "progressive",
            "xyz_encoding_config": {
                "otype": "ProgressiveBandHashGrid", "n_levels": n_levels, "n_features_per_level": 2,
                "log2_hashmap_size": log2_T, "base_resolution": base,
                "per_level_scale": 1.447269237440378, "include_xyz": True,
                "start_level": n_levels, "start_step": 0, "update_steps": 1},
            "mlp_network_config": {
                "otype": "VanillaMLP", "activation": "ReLU", "output_activation": "none",
                "n_neurons": hidden, "n_hidden_layers": 2, "sphere_init": True,
                "sphere_init_radius": 0.5, "weight_norm": True, "precision": precision},
        },
    })


def build_model(dev, args):
    import rise_sdf_amd as R
    torch.manual_seed(0)
    model = R.make("neus", c1_config(hidden=args.hidden, precision=getattr(args, "precision", "fp32"))).to(dev)
    enc = model.geometry.encoding.encoding.encoding
    gen = torch.Generator().manual_seed(0)
    with torch.no_grad():
        enc.params.copy_(((torch.rand(enc.params.numel(), generator=gen) * 2 - 1) * 1e-4).to(dev))
        # The reference's sphere init zeroes the first layer's hash-feature columns; give them small
        # random values so that table gradients are non-trivial (same work either way).
        l0 = model.geometry.network.layers[0]
        l0.weight_v[:, 3:] = (torch.randn(l0.weight_v[:, 3:].shape, generator=gen) * 0.05).to(dev)
    model.train()
    model.geometry.update_step(0, 0)
    model.cos_anneal_ratio = 1.0
    return model


_STREAMS = {}


def side_streams(dev, n):
    """n HIP streams per device, created once (the chunks of a step alternate over them)."""
    key = (str(dev), n)
    if key not in _STREAMS:
        _STREAMS[key] = [torch.cuda.Stream(device=dev) for _ in range(n)]
    return _STREAMS[key]


class HbmWatch:
    """Free device memory sampled from a second thread while a measurement runs: ``report()`` -> GiB free before, at the
    lowest point, and what torch's caching allocator had reserved then (the rest is outside it: code objects, kernel scratch).
    The c1 step allocates per-chunk buffers whose sizes follow the chunk's sample count; the allocator's reserved memory is
    what decides whether two chunks in flight fit the 288 GB."""

    def __init__(self, dev):
        import threading
        self.dev, self.stop = dev, False
        self.free0 = self.lo = torch.cuda.mem_get_info(dev)[0]
        self.reserved = torch.cuda.memory_reserved(dev)
        self.th = threading.Thread(target=self._poll, daemon=True)

    def _poll(self):
        while not self.stop:
            f = torch.cuda.mem_get_info(self.dev)[0]
            if f < self.lo:
                self.lo, self.reserved = f, torch.cuda.memory_reserved(self.dev)
            time.sleep(0.005)

    def __enter__(self):
        self.th.start()
        return self

    def __exit__(self, *exc):
        self.stop = True
        self.th.join()
        return False

    def report(self):
        g = 2.0 ** 30
        return {"free_before": round(self.free0 / g, 1), "min_free": round(self.lo / g, 1),
                "allocator_reserved_then": round(self.reserved / g, 1)}


def run_step(model, rays, jitter, cot, chunk, streams=1):
    """One pass over all rays: fwd+bwd per chunk, gradients accumulated.  Returns the number of samples.

    ``streams`` > 1: consecutive chunks are issued on alternating HIP streams, so that the kernels of two chunks are in
    flight together.  The four large kernels of a chunk are bound by different units (stencil gather: L2 line requests;
    fused MLP forward: vector issue, capped at 192 registers so that a gather wave fits beside two of its waves; MLP
    backward: vector issue + matrix pipe with the whole register file; hash backward: queue traffic and latency), and a
    kernel's tail leaves CUs idle that the other chunk's kernels fill.  Each chunk's whole forward + backward
    stays on ONE stream (autograd runs a node's backward on the stream of its forward); the leaves' ``.grad`` accumulation
    is ordered by autograd's own stream synchronisation, and everything is joined before the gradients are used."""
    n = rays.shape[0]
    total = 0
    if streams <= 1 and os.environ.get("RSDF_BENCH_PREFETCH", "1") == "0":
        for s in range(0, n, chunk):
            e = min(s + chunk, n)
            out = model.forward_(rays[s:e], stratified_u=jitter[s:e])
            total += int(out["ray_indices"].numel())
            torch.autograd.backward([out["opacity"], out["depth"], out["comp_normal_raw"]],
                                    [cot[0][s:e], cot[1][s:e], cot[2][s:e]])
        return total
    if streams <= 1:
        # One chunk in flight, its successor's SAMPLING prefetched: the marcher of chunk i + 1 (two small kernels and the
        # reference's one host read of the sample count, ray_marching.cu:261) is issued on a side stream while chunk i's
        # forward + backward are queued on the main one, so the host read drains the side stream only and the main stream
        # never runs dry between chunks.  Same kernels, same order per chunk, everything inside the timed region; the extra
        # footprint is the next chunk's sample arrays (16 B per sample).
        main = torch.cuda.current_stream()
        side = side_streams(rays.device, 1)[0]

        def sample(s, e):
            with torch.cuda.stream(side):
                with torch.no_grad():
                    ro, rd = rays[s:e, :3].contiguous(), rays[s:e, 3:].contiguous()
                    ri, ts, te = model.occupancy_grid.sampling(ro, rd, render_step_size=model.render_step_size,
                                                               stratified_u=jitter[s:e], cone_angle=0.0, alpha_thre=0.0)
                ev = torch.cuda.Event()
                ev.record(side)
            return ro, rd, ri, ts, te, ev

        side.wait_stream(main)
        bounds = [(s, min(s + chunk, n)) for s in range(0, n, chunk)]
        nxt = sample(*bounds[0])
        for k, (s, e) in enumerate(bounds):
            ro, rd, ri, ts, te, ev = nxt
            main.wait_event(ev)
            for t in (ro, rd, ri, ts, te):
                t.record_stream(main)
            out = model.render_samples(ro, rd, ri, ts, te, e - s)
            total += int(ri.numel())
            torch.autograd.backward([out["opacity"], out["depth"], out["comp_normal_raw"]],
                                    [cot[0][s:e], cot[1][s:e], cot[2][s:e]])
            if k + 1 < len(bounds):
                nxt = sample(*bounds[k + 1])
        return total
    main = torch.cuda.current_stream()
    pool = side_streams(rays.device, streams)
    for st in pool:
        st.wait_stream(main)                       # inputs / zeroed gradients produced on the caller's stream
    for k, s in enumerate(range(0, n, chunk)):
        e = min(s + chunk, n)
        with torch.cuda.stream(pool[k % streams]):
            out = model.forward_(rays[s:e], stratified_u=jitter[s:e])
            total += int(out["ray_indices"].numel())
            torch.autograd.backward([out["opacity"], out["depth"], out["comp_normal_raw"]],
                                    [cot[0][s:e], cot[1][s:e], cot[2][s:e]])
    for st in pool:
        main.wait_stream(st)                       # join: the caller (all-reduce, optimizer) sees complete gradients
    return total


def cpu_baseline(model, rays_cpu, jitter_cpu, max_rays, budget_s=5.0, reps=5):
    """The oracle (a port of the reference path: C + OpenMP for the hash grid and the per-ray loops, torch CPU ops
    for the MLP / elementwise layers) timed on this host's cores, BASELINE.md section 4's protocol: a contiguous
    slice of the same view starting at the middle row, fwd+bwd of field query + NeuS alpha + composite, 1 warm-up +
    ``reps`` (5) timed repetitions, median; the marcher is timed separately.  The slice is ``max_rays`` (4096)
    rays unless a 128-ray calibration pass says that would exceed ``budget_s`` per repetition (the default bench.py run
    must stay within minutes); the line says when it was shrunk."""
    import statistics
    import oracle
    import ctypes
    gomp = ctypes.CDLL("libgomp.so.1")
    n = rays_cpu.shape[0]
    s0 = (n // 2 // 800) * 800
    meta, table, mlp, var = oracle.params_from_model(model)
    roi = torch.tensor([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5])

    def one(n_rays):
        rays = rays_cpu[s0:s0 + n_rays].contiguous()
        u = jitter_cpu[s0:s0 + n_rays].contiguous()
        t0 = time.perf_counter()
        ri, ts, te = oracle.ray_marching(rays[:, :3].contiguous(), rays[:, 3:].contiguous(), scene_aabb=roi,
                                         near_plane=0.0, far_plane=1e10,
                                         render_step_size=model.render_step_size, stratified_u=u)
        t1 = time.perf_counter()
        for q in table, *[t for layer in mlp for t in layer.values()], var:
            if isinstance(q, torch.Tensor) and q.grad is not None:
                q.grad = None
        ref = oracle.neus_geometry_render(rays, ri, ts, te, table, meta, mlp, var, radius=1.5,
                                          fd_eps=model.geometry._finite_difference_eps)
        (ref["opacity"].sum() + ref["depth"].sum() + ref["comp_normal"].sum()).backward()
        t2 = time.perf_counter()
        return int(ri.numel()), t1 - t0, t2 - t1

    # thread count: the node shows every host core but a container's CPU quota can be far smaller, and past it
    # both torch and OpenMP collapse (256 threads: 20x slower than 16 on the round-2 box).  Calibrate on 128 rays.
    one(128)                                  # pages everything in
    cores, dt0 = 1, float("inf")
    for t in (8, 16, 32, 64, 128, 256):
        if t > (os.cpu_count() or 1):
            break
        torch.set_num_threads(t)
        gomp.omp_set_num_threads(t)
        dt = min(one(128)[2] for _ in range(2))
        if dt < dt0:
            cores, dt0 = t, dt
        elif dt > 2.0 * dt0:
            break
    torch.set_num_threads(cores)
    gomp.omp_set_num_threads(cores)
    n_rays = int(max(128, min(max_rays, (budget_s / max(dt0, 1e-6)) * 128) // 128 * 128))
    one(n_rays)                               # warm-up at the timed size
    runs = [one(n_rays) for _ in range(reps)]
    S = runs[0][0]
    dt = statistics.median(r[2] for r in runs)
    dm = statistics.median(r[1] for r in runs)
    return {"value": S / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "slice_shrunk_below_4096_rays": bool(n_rays < max_rays),
            "sample": f"{n_rays} rays" + (f" (shrunk from {max_rays}: ~{budget_s:.0f} s per repetition)" if n_rays < max_rays else "") +
                      f" of the same 800x800 view (pixels {s0}..{s0 + n_rays - 1}), {S} samples; "
                      f"fwd+bwd of field query + alpha + composite: median of {reps} after 1 warm-up = {dt:.2f} s "
                      f"(min {min(r[2] for r in runs):.2f}, max {max(r[2] for r in runs):.2f}); marcher timed "
                      f"separately: {dm * 1e3:.1f} ms ({S / max(dm, 1e-9):.3g} samples/s, 1 thread); hash grid in C "
                      f"with OpenMP, MLP in torch; {cores} threads each = the fastest of 8..{os.cpu_count()} on a "
                      f"128-ray calibration ({os.cpu_count()} cores visible)"}


def roofline_from(summary, steps):
    """Pick the entry point with the most device time and price it against its roofline."""
    def cost(name, a):
        if name in ("rsdf_sdfmlp_fd7_fwd_x2", "rsdf_sdfmlp_fd7_bwd_x2") and a and a[0] == 1:   # parts = 1: the 16-bit form
            b, w = cost(name, (2,) + tuple(a[1:]))
            return "mfma_bf16", w
        if name.endswith("_bf16"):           # config[4]'s bf16 MLP mode: same algorithmic flops, bf16 matrix peak
            b, w = cost(name[:-5], a)
            return ("mfma_bf16" if b == "mfma" else b), w
        # scalar args in ABI order (see include/risesdf_hip.h)
        if name == "rsdf_hashgrid_fwd":      # n, n_active, ld_out, col_off, write_xyz, scale, offset
            n, L = a[0], 16
            return "hbm", n * (L * 8 * 2 * 4 + 12 + L * 2 * 4)
        if name == "rsdf_hashgrid_bwd":      # n, n_active, ld, col_off
            n, L = a[0], 16
            return "hbm", n * (L * 8 * 2 * 4 + 12 + L * 2 * 4)
        if name in ("rsdf_hashgrid_fwd_fd7", "rsdf_hashgrid_bwd_fd7"):   # n_samples, n_active, ...
            return "hbm", 7 * a[0] * (16 * 8 * 2 * 4 + 12 + 16 * 2 * 4)   # 7 evaluations x 1164 B
        if name in ("rsdf_hashgrid_fwd_fd7_pts", "rsdf_hashgrid_bwd_fd7_pts", "rsdf_hashgrid_fwd_fd7_x2"):
            return "hbm", 7 * a[2] * (16 * 8 * 2 * 4 + 12 + 16 * 2 * 4)       # radius, eps, n_samples, n_active, ...
        if name == "rsdf_sdfmlp_fd7_fwd_x2":  # parts, L, H, N2, n_samples
            K0, H, S = 3 + 2 * a[1], a[2], a[4]
            return "mfma", 2.0 * 7 * S * (K0 * H + H * H + H)
        if name == "rsdf_sdfmlp_fd7_bwd_x2":  # parts, L, n_active, H, N2, n_samples
            K0, H, S = 3 + 2 * a[1], a[3], a[5]
            return "mfma", 2 * 2.0 * 7 * S * (K0 * H + H * H + H)
        if name == "rsdf_sdfmlp_fd7_fwd":    # L, n_active, xyz_scale, xyz_offset, H, N2, n_samples
            K0, H, S = 3 + 2 * a[0], a[4], a[6]
            return "mfma", 2.0 * 7 * S * (K0 * H + H * H + H)            # last layer: SDF column only
        if name == "rsdf_sdfmlp_fd7_bwd":    # algorithmic: input gradient + weight gradient = 2x forward
            K0, H, S = 3 + 2 * a[0], a[4], a[6]   # (the in-kernel recompute of the hidden layers is not counted)
            return "mfma", 2 * 2.0 * 7 * S * (K0 * H + H * H + H)
        if name in ("rsdf_pair_fwd16", "rsdf_pair_bwd16"):     # the pair kernels' 16-bit mode: same algorithmic flops, one product
            return cost(name[:-2], a)
        if name == "rsdf_pair_fwd":          # K, n, N2, out_act: two 128-wide layers (+ the folded narrow output layer)
            return "mfma", 2.0 * a[1] * (a[0] * 128 + 128 * 128 + 128 * a[2])
        if name == "rsdf_pair_bwd":          # K, n, g_masked, N2, ...: input + weight gradients = 2x forward (recompute not counted)
            return "mfma", 2 * 2.0 * a[1] * (a[0] * 128 + 128 * 128 + 128 * a[3])
        if name == "rsdf_linear_fwd":        # ldx, n, K, N, act, ldy
            return "mfma", 2.0 * a[1] * a[2] * a[3]
        if name == "rsdf_linear_bwd_input":  # lddy, n, K, N, act, k0, Kout, lddx
            return "mfma", 2.0 * a[1] * a[3] * a[6]
        if name == "rsdf_linear_bwd_weight":  # lddz, ldx, n, K, N
            return "mfma", 2.0 * a[2] * a[3] * a[4]
        return None, 0.0

    def pipe_products(name, a):
        """MFMA products the kernel EXECUTES per algorithmic fp32 product, on the 16-bit matrix pipe it runs on: the x2
        kernels carry every operand as two fp16 parts and evaluate three partial products (one in the 16-bit mode), the
        round 1-3 kernels and the per-layer kernels three bf16 parts and six, the _bf16 build one."""
        if name.endswith("_bf16") or name in ("rsdf_pair_fwd16", "rsdf_pair_bwd16"):
            return 1
        if name in ("rsdf_sdfmlp_fd7_fwd_x2", "rsdf_sdfmlp_fd7_bwd_x2"):
            return 1 if (a and a[0] == 1) else 3
        if name in ("rsdf_pair_fwd", "rsdf_pair_bwd"):     # two fp16 parts, three products (csrc/mlp_pair.hip)
            return 3
        return 6

    def price(name, v):
        """-> the roofline entry of one entry point.  MFMA-bound entries are priced against the pipe they execute on:
        ``achieved`` stays the ALGORITHMIC (fp32-equivalent) TFLOP/s, ``executed`` = achieved x products per fp32 product,
        ``frac`` = executed / the dense f16 / bf16 MFMA peak (a fraction of the fp32-MFMA peak can exceed 1 for a kernel that
        does not run on that pipe, and did).  The stencil gather is priced on the bytes it cannot avoid moving (SURVEY 8d's
        7 x 1164 B describes 56 corner fetches per level; the kernel merges them from an L2 / MALL-resident table)."""
        b, _ = cost(name, v["args"][0])
        if b is None:
            return None
        wk = sum(cost(name, a)[1] for a in v["args"])
        secs = v["ms"] / 1e3
        if b == "hbm":
            e = {"bound": "hbm", "achieved": round(wk / secs / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s"}
            if name in ("rsdf_hashgrid_fwd_fd7_x2", "rsdf_hashgrid_fwd_fd7_pts", "rsdf_hashgrid_fwd_fd7"):
                # centre 12 B + planes 896 B per sample -- the x2 image: 1008 B with its xyz / bias columns
                nb = 1020 if name == "rsdf_hashgrid_fwd_fd7_x2" else 908
                e["survey_8d_GBps"] = e["achieved"]          # 7 evaluations x 1164 B: not a bound for a merged stencil gather
                e["evals_per_sec"] = round(e["achieved"] * 1e9 / 1164.0)
                e["necessary_bytes_per_sample"] = nb
                e["achieved"] = round(e["achieved"] * nb / (7 * 1164.0), 1)
            e["frac"] = round(e["achieved"] / HBM_PEAK_GBS, 4)
            return e
        m = pipe_products(name, v["args"][0])
        ach = wk / secs / 1e12
        return {"bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                "products_per_fp32_product": m, "executed": round(m * ach, 1),
                "frac": round(m * ach / MFMA_BF16_PEAK_TF, 4)}

    def samples_of(name, a):
        if name.endswith("_fd7_pts") or name == "rsdf_hashgrid_fwd_fd7_x2":
            return a[2]
        if name == "rsdf_sdfmlp_fd7_fwd_x2":
            return a[4]
        if name == "rsdf_sdfmlp_fd7_bwd_x2":
            return a[5]
        if name in ("rsdf_pair_fwd", "rsdf_pair_bwd", "rsdf_pair_fwd16", "rsdf_pair_bwd16"):
            return a[1]
        return a[6] if name.startswith("rsdf_sdfmlp_fd7") else a[0]   # (also the _bf16 names)

    best = max(summary.items(), key=lambda kv: kv[1]["ms"])
    name, d = best
    bound, _ = cost(name, d["args"][0])
    breakdown = {k: {"calls": v["calls"], "ms_per_step": round(v["ms"] / steps, 3)}
                 for k, v in sorted(summary.items(), key=lambda kv: -kv[1]["ms"])}
    if bound is None:
        return {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                "traffic": None, "kernel": name}, breakdown
    out = price(name, d)
    out.update({"traffic": None, "kernel": name, "avg_launch_ms": round(d["ms"] / d["calls"], 4), "launches": d["calls"],
                "samples_per_launch": round(sum(samples_of(name, a) for a in d["args"]) / d["calls"])})
    # the same pricing for the other heavy entry points (the north star quotes the hash gather separately)
    others = {}
    for k, v in summary.items():
        if k == name or v["ms"] < 0.02 * d["ms"]:
            continue
        e = price(k, v)
        if e is not None:
            others[k] = e
    out["other_kernels"] = others
    if out["bound"] == "mfma":
        out["note"] = ("achieved = algorithmic fp32-equivalent TFLOP/s; executed = achieved x the matrix products the kernel "
                       "issues per fp32 product (x2: two fp16 parts, 3; round-3 / per-layer: three bf16 parts, 6; 16-bit "
                       "modes: 1); frac = executed / dense f16-bf16 MFMA peak, the pipe the kernel runs on")
    return out, breakdown


def generic_gather_probe(model, rays, jitter, chunk):
    """The generic tcnn-shaped encoder (rsdf_hashgrid_fwd: what every reference-shaped caller of tcnn.Encoding uses) on
    the centre points of one bench chunk (~19 M ray samples), timed with HIP events outside the timed region."""
    from rise_sdf_amd import ops
    with torch.no_grad():
        ro, rd = rays[:chunk, :3].contiguous(), rays[:chunk, 3:].contiguous()
        ri, ts, te = model.occupancy_grid.sampling(ro, rd, render_step_size=model.render_step_size,
                                                   stratified_u=jitter[:chunk], cone_angle=0.0, alpha_thre=0.0)
        x7 = ops.fd_points(ro, rd, ri, ts, te, model.geometry.radius, model.geometry._finite_difference_eps)
        x = x7.view(-1, 7, 3)[:, 0].contiguous()
        del x7
        enc = model.geometry.encoding.encoding.encoding
        enc(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            enc(x)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
    n = x.shape[0]
    gbs = n * 1164 / (ms / 1e3) / 1e9
    return {"bound": "hbm", "evals": n, "avg_launch_ms": round(ms, 3), "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "evals_per_sec": n / (ms / 1e3),
            "note": f"generic per-point gather (rsdf_hashgrid_fwd_staged at this size), 1164 algorithmic B per evaluation, "
                    f"centre points of one {chunk}-ray chunk",
            # the stated bound of this kernel: its planes pass issues 72.6 L2 requests per evaluation
            # (13 hashed levels x 4.5 lines + the dense levels; TCC_REQ, profiles/*/pmc/gather_l2.csv) against the ~2.7e11
            # requests/s the L2s deliver (tools/gather_pair_bench.hip: 2.5e11 measured) -> <= 3.7e9 evaluations/s = 0.54 of
            # the 8 TB/s x 1164 B figure: the north star's 0.60 is out of reach for unsorted points with this hash
            "l2_request_ceiling": {"evals_per_sec": 3.7e9, "frac_of_hbm_roofline": 0.54, "requests_per_eval": 72.6,
                                   "frac_of_ceiling": round(n / (ms / 1e3) / 3.7e9, 3),
                                   "source": "DESIGN.md 4 'The generic gather in two passes'; tools/pmc_gather.sh"}}


def attach_traffic(roof, path):
    """roofline.traffic: HBM bytes per launch of the dominant kernel from the committed PMC summary (separate
    rocprofv3 --pmc passes, FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 note; tools/pmc_summary.py).  The
    counters cannot be collected inside this run, so the figure is per launch of the same kernel at the workload
    the summary names, rescaled by sample count."""
    if roof is None or not os.path.exists(path):
        return
    try:
        pmc = json.load(open(path))
    except Exception:
        return
    entry = pmc.get("kernels", {}).get(roof.get("kernel"))
    if not entry:
        return
    per_sample = entry["hbm_bytes_per_sample"]
    roof["traffic"] = per_sample * roof.get("samples_per_launch", 0) or None
    roof["traffic_build"] = pmc.get("build", "unknown")     # the tree the counters were collected on (tools/pmc_summary.py)
    roof["traffic_source"] = (f"{os.path.relpath(path, ROOT)} ({pmc.get('source', 'separate --pmc passes')}): "
                              f"{per_sample:.0f} B/sample fetched+written x samples per launch")


def measure_c1(model, rays, jitter, cot, chunk, steps, warmup, world=1, buckets=None, timing=True, streams=1):
    """``warmup`` untimed + ``steps`` timed passes over all rays (fwd+bwd per chunk, gradients accumulated, then the
    gradient all-reduce for world > 1) -> dict(dt, samples, summary)."""
    from rise_sdf_amd import _lib

    def step():
        for p in model.parameters():
            p.grad = None
        S = run_step(model, rays, jitter, cot, chunk, streams)
        if buckets is not None:
            buckets.all_reduce_mean(world)
        return S

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    timer = None
    if timing:
        timer = _lib.KernelTimer()
        _lib.set_timer(timer)
    barrier()
    t0 = time.perf_counter()
    samples = 0
    for _ in range(steps):
        samples += step()
    barrier()
    dt = time.perf_counter() - t0
    _lib.set_timer(None)
    return {"dt": dt, "samples": samples, "summary": timer.summary() if timer is not None else None}


def brief(res, steps):
    """samples/s + the dominant kernel's roofline fraction of a secondary measurement."""
    out = {"samples_per_s": res["samples"] / res["dt"], "ms_per_step": res["dt"] / steps * 1e3,
           "samples_per_step": res["samples"] / steps}
    if res.get("summary"):
        roof, _ = roofline_from(res["summary"], steps)
        if roof is not None:
            out["dominant_kernel"] = {k: roof.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac",
                                                               "avg_launch_ms")}
        top = sorted(res["summary"].items(), key=lambda kv: -kv[1]["ms"])[:6]
        out["top_ms_per_step"] = {k: round(v["ms"] / steps, 2) for k, v in top}
    return out


def secondary_measurements(dev, args, rays, jitter, cot):
    """the numbers outside the headline configuration, measured by the same command so that the
    driver's run records them: c1 at 4096-ray chunks (the reference's ray_chunk), c1 at the yaml's MLP width, the
    per-layer drop-in route INTEGRATION.md leads with, config[2] on the full 800x800 view, and the config[3] training
    step at the reference's 262,144-samples operating point.  Untimed for ``value``; each entry is guarded so that a
    failure is recorded instead of losing the headline line."""
    import gc
    from rise_sdf_amd import _lib as _lib_mod
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    extras = {}

    only = [t for t in (getattr(args, "only_extras", "") or "").split(",") if t]

    def guarded(name, fn):
        if only and name not in only:
            return
        gc.collect()
        torch.cuda.synchronize()
        _lib_mod.free_workspaces()
        torch.cuda.empty_cache()
        # HBM footprint of the variant (HbmWatch)
        t0 = time.perf_counter()
        with HbmWatch(dev) as hw:
            try:
                extras[name] = fn()
            except Exception as e:   # noqa: BLE001  (recorded, not swallowed: the key carries the error)
                extras[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if isinstance(extras[name], dict):
            extras[name]["wall_s"] = round(time.perf_counter() - t0, 1)
            extras[name]["hbm_gib"] = hw.report()

    def c1_variant(hidden, chunk, n_rays=None, fused=True, steps=1, warmup=1, precision="fp32", streams=None):
        a = argparse.Namespace(hidden=hidden, precision=precision)
        m = build_model(dev, a)
        if not fused:
            m.config["fused"] = False
        r = rays if n_rays is None else rays[:n_rays]
        ns = args.streams if streams is None else streams
        res = measure_c1(m, r, jitter[:r.shape[0]], [c[:r.shape[0]] for c in cot], chunk, steps, warmup, streams=ns)
        out = brief(res, steps)
        out["config"] = {"hidden": hidden, "chunk_rays": chunk, "hip_streams": ns, "rays": int(r.shape[0]),
                         "mlp_precision": precision,
                         "fused_stencil_kernels": bool(m._fused_ok())}
        return out

    guarded("chunk4096", lambda: c1_variant(args.hidden, 4096, streams=2))          # the reference's ray_chunk; ~31 GiB
    # two chunks in flight on alternating HIP streams (rounds 3-4's default: +1-2 % for twice the footprint)
    guarded("two_streams", lambda: c1_variant(args.hidden, args.chunk, streams=2, steps=3, warmup=2))
    # (H = 128: the cooperative kernels hold every register of a CU, a second chunk in flight only adds contention)
    h128_chunk = min(args.chunk, 24576)       # (the H = 128 variants keep the chunk size they were measured at)
    guarded("h128", lambda: c1_variant(128, h128_chunk, streams=1))
    # config[4]'s opt-in 16-bit MLP modes at the yaml's width (never part of the f32 headline): 'bf16' = the round-3
    # kernels with one bf16 product per k-step; 'fp16' = the x2 kernels with ONE fp16 part (11 significant bits)
    guarded("h128_bf16", lambda: c1_variant(128, h128_chunk, precision="bf16"))
    guarded("h128_fp16", lambda: c1_variant(128, h128_chunk, precision="fp16"))
    guarded("h64_fp16", lambda: c1_variant(64, args.chunk, precision="fp16"))
    # the per-layer API route (tcnn.Encoding / VanillaMLP shaped calls, one or a few kernels each; INTEGRATION.md's
    # two-line dropin.install()): [7 S, 35] rows through HBM, so a quarter of the view at the reference's chunk size
    # (one chunk at a time, as the reference's own loop issues it -- and because with two chunks in flight the per-entry-point
    #  times include the neighbour chunk's share: the round-3 line's "rsdf_linear_bwd_weight 7.3 ms, 0.094" was 2.5 ms alone)
    guarded("dropin_path", lambda: c1_variant(args.hidden, 4096, n_rays=rays.shape[0] // 4 // 800 * 800, fused=False, streams=1))

    def c2(tex_precision="fp32", streams=1):
        from bench_c2 import measure_c2
        # (~9 KB of scratch per sample with the radiance networks' activations: 16384 rays one at a time, 8192 when two
        # chunks are in flight -- two 16384-ray chunks exceed the HBM and the caching allocator thrashes, 1.7e7)
        r = measure_c2(dev, args.width, args.height, 16384 if streams == 1 else 8192, stage=1, tex_hidden=128, steps=1,
                       tex_precision=tex_precision, streams=streams)
        out = brief({"dt": r["ms_per_step"] / 1e3, "samples": r["samples_per_step"], "summary": r["summary"] or None}, 1)
        out["workload"] = r["workload"]
        return out
    guarded("c2_800", c2)                                           # one chunk at a time: carries the dominant kernel
    # (two chunks in flight make c2 slower, 5.2e7: allocator pressure -- measured through round 4, no longer run by default)
    # configs[4]'s bf16 mode on the radiance networks only (SDF network fp32: the combination the convergence proxy,
    # tests/test_gpu_convergence.py, finds indistinguishable from fp32)
    guarded("c2_800_bf16_radiance", lambda: c2("bf16"))

    def c3(**kw):
        from bench_step import measure
        r = measure(dev, stage=1, steps=30, settle=80, syncs=True, **kw)
        return {k: r[k] for k in ("ms_per_step", "ms_per_step_with_entry_point_events", "rays_per_step", "samples_per_step",
                                  "samples_per_s", "rsdf_kernel_ms_per_step", "rsdf_kernel_ms_note", "host_syncs_per_step", "sampler_stats", "top",
                                  "hidden",
                                  "stage")}
    guarded("c3_step", c3)
    guarded("c3_step_bf16_radiance", lambda: c3(tex_precision="bf16"))

    def rccl():
        """RCCL on this box: a one-rank ``nccl`` group in a CHILD process (this one owns no process group
        and must not exec over an initialised GPU) drives the gradient exchange of the N-rank step over the real tensors."""
        import subprocess
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RSDF_DIST_SHARE_GPU", "RSDF_DIST_BACKEND"):
            env.pop(k, None)
        import socket
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        env["MASTER_PORT"] = str(sock.getsockname()[1])
        sock.close()
        r = subprocess.run([sys.executable, "-m", "rise_sdf_amd.dist", "--selftest"], env=env, cwd=ROOT, capture_output=True,
                           text=True, timeout=600)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
        if r.returncode != 0 or not lines:
            raise RuntimeError(f"rccl selftest exit {r.returncode}: {r.stderr[-200:]}")
        out = json.loads(lines[-1][7:])
        out["note"] = "one rank on cuda:0: the collectives execute on the device; values unchanged by construction"
        return out
    guarded("rccl_selftest", rccl)
    return extras


def run_c3(args, rank, local, world, dev):
    """BASELINE.json config[3]: the split-mixed-occ training step with occupancy-grid marching, <= 4096 rays per rank
    steered to 262,144 samples per step (systems/split_occ.py:51,159-161), one process per GPU, gradient mean over the
    ranks after every backward (launch.py:84-97).  A step = update_step (occupancy update every 16th) + ray batch +
    build_mips + forward + loss + backward + all-reduce + Adam; value = surviving samples of all ranks per second."""
    from rise_sdf_amd import _lib
    from rise_sdf_amd.step import build_synthetic_training
    model, ts = build_synthetic_training(dev, stage=1, hidden=args.hidden if args.hidden != 64 else 128, rank=rank,
                                         world=world)
    gs = 20000
    warm = max(args.warmup, 80)                      # dynamic_ray_sampling needs ~60 steps to reach the operating point
    for k in range(warm):
        ts.step(gs + k)
    gs += warm
    timer = None
    if rank == 0 and not args.no_kernel_timing:
        timer = _lib.KernelTimer()
        _lib.set_timer(timer)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    samples = rays = 0
    for k in range(args.steps):
        r = ts.step(gs + k)
        samples += r["num_samples"]
        rays += r["num_rays"]
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _lib.set_timer(None)
    tt = torch.tensor([dt, float(samples), float(rays)], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = tt[0:1].clone()
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        ssum = tt[1:3].clone()
        torch.distributed.all_reduce(ssum, op=torch.distributed.ReduceOp.SUM)
        dt, samples, rays = float(tmax), float(ssum[0]), float(ssum[1])
    if rank == 0:
        roof = breakdown = None
        if timer is not None:
            roof, breakdown = roofline_from(timer.summary(), args.steps)
        print(json.dumps({
            "metric": "ray-marched SDF samples/sec (fwd+bwd), occupancy-pruned training step", "value": samples / dt,
            "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": warm,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "c3: split-mixed-occ (yaml sizes) training step, occupancy-grid marching, <= 4096 rays "
                                   "per rank steered to 262,144 samples, stage 1, grad all-reduce",
                       "hidden": int(model.geometry.network.n_neurons), "rccl_ranks": world,
                       "dist_backend": torch.distributed.get_backend() if world > 1 else None,
                       "rays_per_step": rays / args.steps, "samples_per_step": samples / args.steps,
                       "parallelism": f"ray-parallel x{world}"},
            "roofline": roof, "cpu_baseline": None, "kernel_breakdown": breakdown}))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def _workspace_stats():
    from rise_sdf_amd import _lib
    return _lib.workspace_stats()


def _lib_status_totals(dev):
    """The sticky status counters after the run (rise_sdf_amd._lib.poll_status): forward range violations raise; backward
    launches the range guard rerouted to the range-free kernels are reported."""
    from rise_sdf_amd import _lib
    try:
        return _lib.poll_status(dev)
    except Exception as e:   # noqa: BLE001  (recorded, not swallowed)
        return {"error": str(e)[:200]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c1", choices=["c1", "c3"],
                    help="c1 = BASELINE.json config[1] (the metric); c3 = the occupancy-pruned N-rank training step")
    ap.add_argument("--chunk", type=int, default=28672,
                    help="rays per forward/backward chunk (~5 KB of HBM scratch per sample and chunk in flight: 28672 rays = "
                         "~89 GiB reserved with one chunk in flight)")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams the chunks of a step alternate over (run_step).  Default 1 = one chunk at a time "
                         "(~89 GiB reserved); 2 = two chunks in flight (+0-1 %% throughput for ~174 GiB: the "
                         "`secondary.two_streams` entry of the default run)")
    ap.add_argument("--width", type=int, default=800)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--hidden", type=int, default=64)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "fp16"],
                    help="MLP matrix precision: fp32 = fp32-equivalent split products (the metric's dtype f32); bf16 = "
                         "BASELINE.json configs[4]'s opt-in bf16 MLP mode (reported with dtype bf16)")
    ap.add_argument("--cpu-rays", type=int, default=4096,
                    help="upper bound on the rays in the CPU-baseline sample (0 = skip); shrunk to fit ~8 s/repetition")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--time-overlapped", action="store_true",
                    help="also record per-entry-point HIP events inside a two-stream timed region (overlapping times)")
    ap.add_argument("--only-extras", default="",
                    help="comma-separated names: run only these `secondary` measurements (e.g. dropin_path,c3_step)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary measurements appended to the line at N = 1 (see secondary_measurements)")
    ap.add_argument("--pmc-summary", default=os.path.join(ROOT, "profiles", "pmc_summary.json"),
                    help="committed per-kernel HBM traffic from separate rocprofv3 --pmc passes (tools/pmc_passes.sh)")
    args = ap.parse_args()

    from rise_sdf_amd import dist as rdist
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: launch the N ranks ourselves.  Nothing in this process has touched the
        # GPU yet (torch.cuda.device_count() does not initialise it), and the children are plain subprocesses.
        n_dev = torch.cuda.device_count()
        if n_dev < args.gpus and os.environ.get("RSDF_DIST_SHARE_GPU") != "1":
            sys.exit(f"bench.py: --gpus {args.gpus} but only {n_dev} GPU(s) are visible")
        sys.exit(rdist.spawn_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus))
    rank, local, world = rdist.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback for the product path)"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if args.workload == "c3":
        return run_c3(args, rank, local, world, dev)

    from rise_sdf_amd.ray_utils import orbit_view_rays
    model = build_model(dev, args)
    rays = orbit_view_rays(args.width, args.height, seed=rdist.rank_seed(0, rank), device=dev)   # HIP ray generator
    n_rays = rays.shape[0]
    g = torch.Generator().manual_seed(2 + rank)
    jitter_cpu = torch.rand(n_rays, generator=g)
    rays_cpu, jitter = rays.cpu(), jitter_cpu.to(dev)
    cot = [torch.randn(n_rays, 1, generator=g).to(dev), torch.randn(n_rays, 1, generator=g).to(dev),
           torch.randn(n_rays, 3, generator=g).to(dev)]
    buckets = rdist.GradBuckets(model.parameters())

    # per-entry-point HIP events inside the timed region only when the step runs on one stream: with two chunks in flight
    # those times overlap and are not used (the roofline object then comes from a one-stream pass below)
    with HbmWatch(dev) as hbm_watch:
        res = measure_c1(model, rays, jitter, cot, args.chunk, args.steps, args.warmup, world, buckets,
                         timing=(rank == 0 and not args.no_kernel_timing and (args.streams == 1 or args.time_overlapped)),
                         streams=args.streams)
    dt, samples = res["dt"], res["samples"]

    tt = torch.tensor([dt, float(samples)], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = tt[0:1].clone()
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        ssum = tt[1:2].clone()
        torch.distributed.all_reduce(ssum, op=torch.distributed.ReduceOp.SUM)
        dt, samples = float(tmax), float(ssum)

    if rank == 0:
        roof, breakdown, overlapped = (None, None, None)
        summary, n_timed = res["summary"], args.steps
        if args.streams > 1 and not args.no_kernel_timing:
            # With two chunks in flight a kernel's event-to-event time includes its neighbour's share of the GPU (the
            # per-entry-point times of the timed region add up to ~1.8x the step), so they cannot price a kernel against
            # its roofline.  The roofline object therefore comes from ONE more pass of the same step issued on one stream,
            # right here in the same run; the timed region's own (overlapped) per-entry-point times are kept beside it.
            if summary is not None:
                _, overlapped = roofline_from(summary, args.steps)
            iso = measure_c1(model, rays, jitter, cot, args.chunk, 1, 0, 1, None, timing=True, streams=1)
            summary, n_timed = iso["summary"], 1
        if summary is not None:
            roof, breakdown = roofline_from(summary, n_timed)
            if roof is not None and args.streams > 1:
                roof["measured_in"] = ("one-stream pass of the same step (1 step, same process, right after the timed "
                                       "region): kernels alone on the GPU; rocprofv3 summary of `bench.py --streams 1` under "
                                       "profiles/")
                k = roof.get("kernel")
                if overlapped and k in overlapped:
                    roof["avg_launch_ms_in_timed_region"] = round(overlapped[k]["ms_per_step"] * args.steps
                                                                  / max(overlapped[k]["calls"], 1), 4)
            attach_traffic(roof, args.pmc_summary)
            if roof is not None and "other_kernels" in roof:
                roof["other_kernels"]["rsdf_hashgrid_fwd (generic)"] = generic_gather_probe(model, rays, jitter, args.chunk)
        cpu = None
        t_cpu0 = time.perf_counter()
        if args.cpu_rays > 0 and world == 1:   # reported at N = 1 only
            cpu = cpu_baseline(model, rays_cpu, jitter_cpu, args.cpu_rays)
        t_cpu = time.perf_counter() - t_cpu0
        line = {
            "metric": "ray-marched SDF samples/sec (fwd+bwd), 800x800 rays, L=16 hashgrid",
            "value": samples / dt, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": {"bf16": "bf16", "fp16": "f16"}.get(args.precision, "f32"),
            "data": "synthetic",
            "config": {"workload": "c1: toaster-sized 800x800 view per GPU, dense marching, L=16 T=2^19 "
                                   f"hash grid + 2x{args.hidden} SDF MLP (7 FD taps), NeuS alpha + composite, fwd+bwd",
                       "hidden": args.hidden, "mlp_precision": args.precision,
                       "fused_stencil_kernels": bool(model._fused_ok()),
                       "rccl_ranks": torch.distributed.get_world_size() if world > 1 else 1,
                       "dist_backend": torch.distributed.get_backend() if world > 1 else None,
                       "rays_per_gpu": n_rays, "chunk_rays": args.chunk, "hip_streams": args.streams,
                       # the number formats behind dtype f32: two-part fp16 operands with range guards;
                       # hash-backward queue records with 20 significant bits; range-guard activity during this run
                       "sdf_mlp_format": ("x2: two fp16 parts per fp32 operand, 3 MFMA products, fp32 accumulate; forward range "
                                          "and backward dynamic range guarded on the device" if args.precision == "fp32" and
                                          os.environ.get("RSDF_X2", "1") != "0" else args.precision),
                       "hash_bwd_records": "20-bit block-float pairs (8 bytes), fp64 reduction",
                       "x2_guard": _lib_status_totals(dev), "workspace_gib": _workspace_stats(),
                       "samples_per_step": samples / args.steps, "hbm_gib": hbm_watch.report(),
                       "field_evals_per_sec": 7 * samples / dt, "parallelism": f"ray-parallel x{world}"},
            "roofline": roof, "cpu_baseline": cpu, "kernel_breakdown": breakdown,
        }
        if overlapped is not None:
            line["kernel_breakdown_timed_region_overlapped"] = overlapped
        line["wall_s"] = {"cpu_baseline": round(t_cpu, 1), "until_secondary": round(time.perf_counter() - T_PROCESS_START, 1)}
        if world == 1 and not args.no_extras:
            del model, buckets
            line["secondary"] = secondary_measurements(dev, args, rays, jitter, cot)
            line["wall_s"]["total"] = round(time.perf_counter() - T_PROCESS_START, 1)
        print(json.dumps(line))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
