"""config[2] measurement: c1 + radiance branch + split-sum shading (stage 1) on one MI355X, fwd+bwd.
Not the bench.py metric (that is config[1]); prints samples/s and the per-entry-point breakdown.
Usage: python tools/bench_c2.py [--width 800 --height 800 --chunk 8192 --stage 1 --tex-hidden 128]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure_c2(dev, width=800, height=800, chunk=16384, stage=1, tex_hidden=128, steps=1, hidden=64, tex_precision="fp32",
               streams=1):
    """-> dict(samples_per_s, ms_per_step, samples_per_step, kernel_ms_total, top, summary): one view of config[2]
    (c1 + radiance branch + split-sum shading against a 512^2 environment), fwd+bwd, one build_mips per step."""
    import types
    args = types.SimpleNamespace(width=width, height=height, chunk=chunk, stage=stage, tex_hidden=tex_hidden, steps=steps)
    import rise_sdf_amd as R
    from rise_sdf_amd import _lib
    from rise_sdf_amd.ray_utils import orbit_view_rays
    import bench
    cfg = bench.c1_config(hidden=hidden)
    mlp = lambda n: {"otype": "VanillaMLP", "activation": "ReLU", "output_activation": "none",
                     "n_neurons": args.tex_hidden, "n_hidden_layers": n, "precision": tex_precision}
    feat = cfg["geometry"]["feature_dim"]
    cfg.update({
        "name": "split-mixed-occ", "indirect_pred": False, "curvature": False,
        "split_sum_kick_in_step": 0 if args.stage else 1 << 60,
        "texture": {"name": "volume-mixed-mip-split-occ", "input_feature_dim": feat, "other_dim": 3, "sample_size": 8,
                    "dir_encoding_config": {"otype": "SphericalHarmonics", "degree": 5, "reflected": True},
                    "metallic_mlp_network_config": mlp(2), "albedo_mlp_network_config": mlp(4),
                    "spec_mlp_network_config": mlp(4), "roughness_mlp_network_config": mlp(2),
                    "secondary_mlp_network_config": mlp(4),
                    "xyz_encoding_config": {"otype": "VanillaFrequency", "n_frequencies": 6},
                    "color_activation": "sigmoid"},
        "light": {"name": "envlight-mip-cube",
                  "envlight_config": {"scale": 0.5, "bias": 0.25, "base_res": 512, "hdr_filepath": None}},
    })
    torch.manual_seed(0)
    model = R.make("split-mixed-occ", R.Config(cfg)).to(dev)
    model.train()
    model.grid_prune = False
    model.occupancy_grid.binaries.fill_(True)
    with torch.no_grad():
        l0 = model.geometry.network.layers[0]
        l0.weight_v[:, 3:] = torch.randn_like(l0.weight_v[:, 3:]) * 0.05
    model.update_step(0, 0)
    model.background_color = torch.ones(3, device=dev)
    rays = orbit_view_rays(args.width, args.height, seed=0, device=dev)
    n = rays.shape[0]
    g = torch.Generator().manual_seed(2)
    u = torch.rand(n, generator=g).to(dev)
    cot = torch.randn(n, 3, generator=g).to(dev)

    def step():
        for p in model.parameters():
            p.grad = None
        em = model.emitter
        if model.stage:
            # one prefilter per step; the chunks see detached leaves of the filtered maps and their gradients are
            # pushed through build_mips once at the end (instead of once per chunk via retain_graph)
            em.build_mips()
            built = em.specular + [em.diffuse]
            leaves = [t.detach().requires_grad_(True) for t in built]
            em.specular, em.diffuse = leaves[:-1], leaves[-1]
        total = 0
        key = "comp_rgb_phys_full" if model.stage else "comp_rgb_full"
        main = torch.cuda.current_stream()
        pool = bench.side_streams(dev, streams) if streams > 1 else [main]   # chunks alternate over the streams (bench.run_step)
        for st in pool:
            st.wait_stream(main)
        for k, s in enumerate(range(0, n, args.chunk)):
            with torch.cuda.stream(pool[k % len(pool)]):
                out = model.forward_(rays[s:s + args.chunk], stratified_u=u[s:s + args.chunk])
                total += out["num_samples_host"]
                (out[key] * cot[s:s + args.chunk]).sum().backward()
        for st in pool:
            main.wait_stream(st)
        if model.stage:
            torch.autograd.backward(built, [l.grad if l.grad is not None else torch.zeros_like(l) for l in leaves])
        return total

    step()
    timer = _lib.KernelTimer()
    if streams == 1:                  # (per-entry-point events only when one chunk is in flight: bench.py, roofline)
        _lib.set_timer(timer)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    samples = sum(step() for _ in range(args.steps))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _lib.set_timer(None)
    summ = timer.summary()
    top = {k: round(v["ms"] / args.steps, 1) for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:14]}
    return {"workload": f"c2 stage {model.stage}: split-mixed-occ, {args.width}x{args.height}, texture width "
                        f"{args.tex_hidden} ({tex_precision}), env 512^2, {streams} stream(s)", "samples_per_s": samples / dt,
            "ms_per_step": dt / args.steps * 1e3, "samples_per_step": samples / args.steps,
            "kernel_ms_total": round(sum(v["ms"] for v in summ.values()) / args.steps, 1), "top": top, "summary": summ}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=800)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--chunk", type=int, default=16384)
    ap.add_argument("--stage", type=int, default=1)
    ap.add_argument("--tex-hidden", type=int, default=128)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--streams", type=int, default=1)
    ap.add_argument("--tex-precision", default="fp32")
    args = ap.parse_args()
    r = measure_c2(torch.device("cuda", 0), args.width, args.height, args.chunk, args.stage, args.tex_hidden, args.steps,
                   tex_precision=args.tex_precision, streams=args.streams)
    r.pop("summary")
    print(json.dumps(r))


if __name__ == "__main__":
    main()
