"""How close is the split-bf16 fused SDF MLP to an fp64 evaluation, next to plain fp32 torch on the same weights?
Prints max / rms absolute SDF errors over random stencil points (H = 64, L = 16 grid, sphere-initialised MLP with
random hash-feature weights).  Run on the GPU box: python tools/accuracy_split_bf16.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import bench
    import rise_sdf_amd as R
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = R.make("neus", bench.c1_config(hidden=64)).to(dev)
    geo = model.geometry
    with torch.no_grad():
        geo.encoding.encoding.encoding.params.uniform_(-1e-1, 1e-1)
        l0 = geo.network.layers[0]
        l0.weight_v[:, 3:] = torch.randn_like(l0.weight_v[:, 3:]) * 0.3
    model.train()
    geo.update_step(0, 20000)
    S = 200000
    g = torch.Generator().manual_seed(1)
    rays_o = torch.zeros(1, 3)
    rays_d = torch.nn.functional.normalize(torch.randn(1, 3, generator=g), dim=-1)
    pts = (torch.rand(S, 3, generator=g) * 2 - 1) * 1.4
    # field through the reference-shaped path (per-layer kernels) gives the hash features; redo the MLP three ways
    from rise_sdf_amd import ops
    x7 = ops.fd_taps(pts.to(dev), geo.radius, geo._finite_difference_eps)            # [S,7,3]
    enc = geo.encoding(x7.view(-1, 3), fd7_eps_unit=geo._eps_unit()).detach()        # [7S,35] fp32 (bit-exact gather)
    wb = [(w.detach(), b.detach()) for w, b in geo.network.effective_weights()]
    def mlp(x, dtype):
        h = x.to(dtype)
        for i, (w, b) in enumerate(wb):
            h = torch.nn.functional.linear(h, w.to(dtype), b.to(dtype))
            if i < len(wb) - 1:
                h = torch.nn.functional.softplus(h, beta=100)
        return h[:, 0]
    ref = mlp(enc, torch.float64)
    t32 = mlp(enc, torch.float32).double()
    per_layer = geo.network(enc)[:, 0].double()
    # fused stencil path on the same points
    ri = torch.zeros(S, dtype=torch.int64, device=dev)
    from rise_sdf_amd import fused
    grid, n_active = geo.encoding._hash()
    x7t = x7.permute(1, 0, 2).contiguous()
    sdf7t, _ = fused.sdf_field_fd7(x7t, grid.params, geo.network.effective_weights(), grid.meta,
                                   grid.n_levels if n_active is None else n_active, geo.encoding.xyz_scale,
                                   geo.encoding.xyz_offset, geo._eps_unit(), False)
    fused_sdf = sdf7t.detach().t().reshape(-1).double()
    scale = float(ref.abs().max())
    for name, v in (("torch fp32 (aten GEMM)", t32), ("per-layer split-bf16 kernels", per_layer),
                    ("fused split-bf16 kernel", fused_sdf)):
        e = (v - ref).abs()
        print(f"{name:32s} max |err| {float(e.max()):.3e}   rms {float(e.pow(2).mean().sqrt()):.3e}   (|sdf| max {scale:.3f})")
    eps = geo._finite_difference_eps
    def fd_grad(s):
        s = s.view(-1, 7)
        return 0.5 * (s[:, 1::2] - s[:, 2::2]) / eps
    gref = fd_grad(ref)
    for name, v in (("torch fp32", t32), ("per-layer split-bf16", per_layer), ("fused split-bf16", fused_sdf)):
        print(f"FD normal error  {name:24s} max {float((fd_grad(v) - gref).abs().max()):.3e}  (eps = {eps:.3e})")


if __name__ == "__main__":
    main()
