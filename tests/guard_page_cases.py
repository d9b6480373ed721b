"""Entry points of the C ABI run on inputs and outputs that end flush against unmapped pages (tests/guard_pages.py).

    python tests/guard_page_cases.py <case>        # exit code 0 = no access past any buffer's end

Each case prints the size it is about to run (flushed), so the last line of a faulting run names the launch.
tests/test_gpu_guard_pages.py runs every case in a child process.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from guard_pages import guarded, guarded_like      # noqa: E402
from rise_sdf_amd import _lib as L                  # noqa: E402
from rise_sdf_amd._lib import check, ptr            # noqa: E402

DEV = torch.device("cuda:0")
# ragged tails of every tile size in the tree (32, 64, 128, 256 rows) plus the size of the launch that faulted in round 5
SIZES = [1, 2, 31, 32, 33, 63, 64, 65, 95, 96, 97, 127, 128, 129, 255, 257, 396, 1000, 4099]


def say(*a):
    print(*a, flush=True)


def _weights(K0, H, N2, g):
    shapes = [(H, K0), (H,), (H, H), (H,), (N2, H), (N2,)]
    return [guarded_like((torch.randn(s, generator=g) * 0.2).to(DEV)) for s in shapes]


def sdf_fwd(H, precision="fp32", env=None):
    """rsdf_sdfmlp_fd7_fwd: x7t [7][S][3], planes [L][7][S][2] -> sdf7t [7][S], feature [S][N2], h2c [S][H]."""
    g = torch.Generator().manual_seed(1)
    for k, v in (env or {}).items():
        os.environ[k] = v
    for Lv in (4, 16):
        K0, N2 = 3 + 2 * Lv, 13
        ws = _weights(K0, H, N2, g)
        for S in SIZES:
            for feature in (False, True):
                say(f"sdf_fwd H={H} L={Lv} S={S} feature={feature}")
                x7t = guarded_like(torch.rand((7, S, 3), generator=g).to(DEV))
                planes = guarded_like(torch.randn((Lv, 7, S, 2), generator=g).to(DEV) * 0.1)
                sdf = guarded((7, S))
                feat = guarded((S, N2)) if feature else None
                h2c = guarded((S, H)) if feature else None
                check(L.mlp_fn("rsdf_sdfmlp_fd7_fwd", precision)(
                    ptr(x7t), ptr(planes), Lv, Lv, 2.0, -1.0, H, N2, *[ptr(w) for w in ws], S, ptr(sdf), ptr(feat),
                    ptr(h2c), None), "sdfmlp_fd7_fwd")
                torch.cuda.synchronize()
                assert torch.isfinite(sdf).all()


def sdf_bwd(H, precision="fp32", env=None):
    """rsdf_sdfmlp_fd7_bwd with every operand guarded."""
    g = torch.Generator().manual_seed(2)
    for k, v in (env or {}).items():
        os.environ[k] = v
    for Lv in (4, 16):
        K0, N2 = 3 + 2 * Lv, 13
        ws = _weights(K0, H, N2, g)
        for S in SIZES:
            for feature in (False, True):
                say(f"sdf_bwd H={H} L={Lv} S={S} feature={feature}")
                x7t = guarded_like(torch.rand((7, S, 3), generator=g).to(DEV))
                planes = guarded_like(torch.randn((Lv, 7, S, 2), generator=g).to(DEV) * 0.1)
                d_sdf = guarded_like(torch.randn((7, S), generator=g).to(DEV))
                d_feat = guarded_like(torch.randn((S, N2), generator=g).to(DEV)) if feature else None
                dh2c = guarded((S, H)) if feature else None
                d_planes = guarded((Lv, 7, S, 2), fill=0.0)
                dws = [guarded(w.shape, fill=0.0) for w in ws]
                check(L.mlp_fn("rsdf_sdfmlp_fd7_bwd", precision)(
                    ptr(x7t), ptr(planes), Lv, Lv, 2.0, -1.0, H, N2, *[ptr(w) for w in ws], S, ptr(d_sdf), ptr(d_feat),
                    ptr(dh2c), ptr(d_planes), *[ptr(w) for w in dws], None), "sdfmlp_fd7_bwd")
                torch.cuda.synchronize()
                assert torch.isfinite(d_planes).all()


def _guard_every_allocation():
    """Installs tests/guard_alloc.cpp for this process: inputs, outputs, saved tensors and scratch of every kernel end flush
    against an unmapped page (must run before the first device allocation)."""
    here = os.path.dirname(os.path.abspath(__file__))
    so = os.path.join(here, "_guard_alloc.so")
    if not os.path.exists(so):      # normally built by __graft_entry__.build(); host code only, a second on any box with hipcc
        import subprocess
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", "-shared", "-fPIC", "-w",
                               os.path.join(here, "guard_alloc.cpp"), "-o", so])
    torch.cuda.memory.change_current_allocator(torch.cuda.memory.CUDAPluggableAllocator(so, "guard_malloc", "guard_free"))


RAGGED_VIEWS = [(1, 1), (3, 5), (8, 8), (17, 13), (33, 31), (64, 50), (96, 67)]


def model_c1(hidden):
    """The c1 model (L = 16, T = 2^19, 2 x ``hidden`` SDF network: the x2 kernels, stencil gather, hash backward, alpha, C1,
    accumulation) forward + backward on views whose ray and sample counts are not multiples of any tile size."""
    import types
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from rise_sdf_amd.ray_utils import orbit_view_rays
    model = bench.build_model(DEV, types.SimpleNamespace(hidden=hidden, precision="fp32"))
    g = torch.Generator().manual_seed(3)
    for w, h in RAGGED_VIEWS:
        rays = orbit_view_rays(w, h, seed=w, device=DEV)
        n = rays.shape[0]
        u = torch.rand(n, generator=g).to(DEV)
        for p in model.parameters():
            p.grad = None
        out = model.forward_(rays, stratified_u=u)
        say(f"model_c1 H={hidden} view {w}x{h}: {n} rays, {int(out['ray_indices'].numel())} samples")
        torch.autograd.backward([out["opacity"], out["depth"], out["comp_normal_raw"]],
                                [torch.randn(n, 1, generator=g).to(DEV), torch.randn(n, 1, generator=g).to(DEV),
                                 torch.randn(n, 3, generator=g).to(DEV)])
        torch.cuda.synchronize()
        assert all(bool(torch.isfinite(p.grad).all()) for p in model.parameters() if p.grad is not None)


def model_c2():
    """config[2] (split-sum model: + the layer-pair radiance networks, texture chain, environment prefilter and lookups)
    forward + backward on ragged views, chunked so that the last chunk is a ragged tail."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench_c2
    for w, h, chunk in [(3, 5, 16384), (17, 13, 100), (40, 30, 777)]:
        say(f"model_c2 view {w}x{h} chunk {chunk}")
        r = bench_c2.measure_c2(DEV, width=w, height=h, chunk=chunk, steps=1)
        say(f"   {r['samples_per_step']:.0f} samples")


CASES = {
    "sdf_fwd_h32": lambda: sdf_fwd(32),
    "sdf_fwd_h64": lambda: sdf_fwd(64),
    "sdf_fwd_h128": lambda: sdf_fwd(128),
    "sdf_fwd_h32_coop": lambda: sdf_fwd(32, env={"RSDF_MLP_FWD": "coop"}),
    "sdf_fwd_h64_bf16": lambda: sdf_fwd(64, "bf16"),
    "sdf_bwd_h32": lambda: sdf_bwd(32),
    "sdf_bwd_h64": lambda: sdf_bwd(64),
    "sdf_bwd_h128": lambda: sdf_bwd(128),
    "sdf_bwd_h64_legacy": lambda: sdf_bwd(64, env={"RSDF_MLP_BWD": "legacy"}),
    "model_c1_h64": lambda: model_c1(64),
    "model_c1_h128": lambda: model_c1(128),
    "model_c2": model_c2,
}


if __name__ == "__main__":
    name = sys.argv[1]
    if name == "--list":
        print("\n".join(CASES))
        sys.exit(0)
    if name.startswith("model_"):
        _guard_every_allocation()
    CASES[name]()
    say(f"{name}: ok")
