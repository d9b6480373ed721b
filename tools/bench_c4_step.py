"""Latency of one config[4]-shaped training step (full split-mixed-occ model, yaml sizes) at a given ray batch.
Usage: python tools/bench_c4_step.py [--rays 4096] [--steps 20] [--stage 1]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--stage", type=int, default=1)
    ap.add_argument("--profile", action="store_true", help="cProfile of the host side (no kernel timer)")
    ap.add_argument("--aten-sites", action="store_true",
                    help="which source lines of the package issue the aten kernels of one step (torch profiler stacks)")
    args = ap.parse_args()
    import rise_sdf_amd as R
    from rise_sdf_amd import _lib, ops
    from rise_sdf_amd.loss import loss_tail
    from rise_sdf_amd.ray_utils import orbit_view_rays
    from test_gpu_c4 import c4_config
    from helpers import sphere_binary
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    cfg = c4_config()
    cfg["split_sum_kick_in_step"] = 0 if args.stage else 1 << 60
    model = R.make("split-mixed-occ", cfg).to(dev)
    model.train()
    model.occupancy_grid.binaries = sphere_binary(128, 0.35, 0.65).to(dev)[None]   # a shell around the r = 0.5 sphere
    model.grid_prune = False                                                     # keep that grid fixed
    with torch.no_grad():
        model.variance.variance.fill_(0.5)
    model.update_step(0, 20000)
    model.background_color = torch.ones(3, device=dev)
    view = orbit_view_rays(800, 800, seed=0, device=dev)
    g = torch.Generator().manual_seed(0)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, eps=1e-12)
    lambdas = {"lambda_rgb_mse": 10.0, "lambda_rgb_phys_mse": 10.0, "lambda_mask": 0.1, "lambda_eikonal": 0.05,
               "lambda_sparsity": 0.01, "lambda_curvature": 1.0}

    def step():
        idx = torch.randint(0, view.shape[0], (args.rays,), generator=g).to(dev)
        rays = view[idx]
        rgb, fg = torch.rand(args.rays, 3, device=dev), torch.ones(args.rays, device=dev)
        if model.stage:
            model.emitter.build_mips()
        out = model(rays)
        loss, _ = loss_tail(out, {"rgb": rgb, "fg_mask": fg}, lambdas)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return int(out["num_samples"])

    for _ in range(3):
        step()
    if args.aten_sites:
        import collections
        import traceback
        from torch.utils._python_dispatch import TorchDispatchMode
        sites = collections.Counter()
        skip = ("aten::empty", "aten::view", "aten::_unsafe_view", "aten::reshape", "aten::detach", "aten::as_strided", "aten::slice",
                "aten::select", "aten::expand", "aten::t", "aten::transpose", "aten::permute", "aten::unsqueeze", "aten::squeeze",
                "aten::alias", "aten::_local_scalar_dense", "aten::empty_like", "aten::empty_strided", "aten::split",
                "aten::unbind", "aten::split_with_sizes", "aten::lift_fresh", "aten::_to_copy", "aten::new_empty")

        class Mode(TorchDispatchMode):
            def __torch_dispatch__(self, func, types, args=(), kwargs=None):
                name = func._schema.name
                if not name.startswith(skip):
                    fr = [f for f in traceback.extract_stack() if "rise_sdf_amd" in f.filename]
                    f = fr[-1] if fr else None
                    where = "%s:%d %s" % (os.path.basename(f.filename), f.lineno, f.name) if f else "?"
                    sites[(where, name)] += 1
                return func(*args, **(kwargs or {}))

        with Mode():
            step()
        print("aten ops by issuing line, forward + Python-side backward of one step: total", sum(sites.values()))
        for (site, name), c in sites.most_common(90):
            print("%4d  %-26s %s" % (c, name, site))
        return
    if args.profile:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pr.enable()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        pr.disable()
        print("ms per step (under cProfile): %.2f" % ((time.perf_counter() - t0) / args.steps * 1e3))
        st = pstats.Stats(pr)
        st.sort_stats("tottime").print_stats(28)
        st.print_callers("item")
        return
    timer = _lib.KernelTimer()
    _lib.set_timer(timer)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    samples = sum(step() for _ in range(args.steps))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _lib.set_timer(None)
    summ = timer.summary()
    top = {k: round(v["ms"] / args.steps, 2) for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:12]}
    print(json.dumps({"rays": args.rays, "stage": model.stage, "ms_per_step": dt / args.steps * 1e3,
                      "steps_per_s": args.steps / dt, "samples_per_step": samples / args.steps,
                      "rsdf_kernel_ms_per_step": round(sum(v["ms"] for v in summ.values()) / args.steps, 2),
                      "top": top}))


if __name__ == "__main__":
    main()
