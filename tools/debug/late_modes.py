"""Debug: late-regime gradient agreement with the oracle for the three backward routes (x2, x2 forced reroute, RSDF_X2=0)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle, bench
import rise_sdf_amd as R
from helpers import camera_rays, rel_err
from test_gpu_model import hip_sdf7, oracle_params
dev = torch.device("cuda:0")
hidden = int(sys.argv[1]) if len(sys.argv) > 1 else 64
variance = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6

def build():
    torch.manual_seed(0)
    cfg = bench.c1_config(hidden=hidden); cfg["num_samples_per_ray"] = 256
    model = R.make("neus", cfg).to(dev)
    enc = model.geometry.encoding.encoding.encoding
    gen = torch.Generator().manual_seed(0)
    with torch.no_grad():
        enc.params.copy_(((torch.rand(enc.params.numel(), generator=gen) * 2 - 1) * 3e-2).to(dev))
        l0 = model.geometry.network.layers[0]
        l0.weight_v[:, 3:] = (torch.randn(l0.weight_v[:, 3:].shape, generator=gen) * 0.3).to(dev)
        model.variance.variance.fill_(variance)
        model.geometry.network.layers[-1].bias[0] += 0.25
    model.train(); model.geometry.update_step(0, 0); model.cos_anneal_ratio = 1.0
    return model, enc

rays = camera_rays(48, 48, seed=21)
u = torch.rand(rays.shape[0], generator=torch.Generator().manual_seed(22))
roi = torch.tensor([-1.5]*3 + [1.5]*3)
g = torch.Generator().manual_seed(23)
go, gd = torch.randn(rays.shape[0], 1, generator=g), torch.randn(rays.shape[0], 1, generator=g)
cache = {}
for mode, env in (("x2", {}), ("x2+forced reroute", {"RSDF_X2_REROUTE": "force"}), ("round-3 kernels", {"RSDF_X2": "0"})):
    for k in ("RSDF_X2", "RSDF_X2_REROUTE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    model, enc = build()
    out = model.forward_(rays.to(dev), stratified_u=u.to(dev))
    ri, ts, te = out["ray_indices"].cpu(), None, None
    ri, ts, te = oracle.ray_marching(rays[:, :3].contiguous(), rays[:, 3:].contiguous(), scene_aabb=roi, near_plane=0.0,
                                     far_plane=1e10, render_step_size=model.render_step_size, stratified_u=u)
    sdf7 = hip_sdf7(model, rays, ri, ts, te)
    key = "r3" if "RSDF_X2" in env else "x2"
    if key not in cache:
        meta2, table2, mlp2, var2 = oracle_params(model)
        ref = oracle.neus_geometry_render(rays, ri, ts, te, table2, meta2, mlp2, var2, radius=1.5,
                                          fd_eps=model.geometry._finite_difference_eps, sdf7_given=sdf7)
        ((ref["opacity"] * go).sum() + (ref["depth"] * gd).sum()).backward()
        cache[key] = (table2, mlp2, var2, ref)
    table2, mlp2, var2, ref = cache[key]
    ((out["opacity"] * go.to(dev)).sum() + (out["depth"] * gd.to(dev)).sum()).backward()
    torch.cuda.synchronize()
    lin = [m for m in model.geometry.network.layers if isinstance(m, torch.nn.Linear)]
    errs = {}
    for i, (m, p) in enumerate(zip(lin, mlp2)):
        for name, k2 in (("weight_v", "v"), ("weight_g", "g"), ("bias", "b")):
            errs[f"{i}.{name}"] = rel_err(getattr(m, name).grad.cpu(), p[k2].grad)
    errs["variance"] = rel_err(model.variance.variance.grad.reshape(1).cpu(), var2.grad.reshape(1))
    gt = enc.params.grad.cpu()
    errs["table"] = float((gt - table2.grad).abs().max() / table2.grad.abs().max())
    errs["opacity"] = rel_err(out["opacity"].cpu(), ref["opacity"].detach())
    errs["depth"] = rel_err(out["depth"].cpu(), ref["depth"].detach())
    print(f"H {hidden} var {variance} [{mode}]: " + ", ".join(f"{k} {v:.2e}" for k, v in errs.items()), flush=True)
    from rise_sdf_amd import _lib
    print("   status", _lib.poll_status(dev, raise_on_error=False), flush=True)
