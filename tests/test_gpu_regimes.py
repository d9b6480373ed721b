"""Regimes that round 1 exercised only through bench.py (which checks nothing):
  * config[1] composed at its REAL field parameters (L = 16, T = 2^19, base 32, 2x64, fused kernels) against the oracle;
  * the hash-grid backward at the size of one 32768-ray bench chunk (~19 M samples): stencil path (queues + LDS
    reduction) vs the generic atomic kernel on the same gradient;
  * the queue-overflow path of the stencil backward (records that find their queue full; reducer count clamp)."""
import ctypes
import os
import subprocess
import sys

import pytest
import torch

import oracle
from helpers import camera_rays, rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("hidden", [64, 128])
def test_c1_real_field_parameters_vs_oracle(dev, hidden):
    """bench.py's c1 model (16 levels, 2^19 entries, base 32, per_level_scale 1.447, 48 features, fused stencil kernels)
    on 2304 rays of a view of the box.  The marching step is 4x the bench's so that the CPU oracle finishes in seconds;
    every kernel runs with the level tables, eps (one finest cell) and widths of the benchmark."""
    import rise_sdf_amd as R
    sys.path.insert(0, ROOT)
    import bench
    torch.manual_seed(0)
    cfg = bench.c1_config(hidden=hidden)
    cfg["num_samples_per_ray"] = 256
    model = R.make("neus", cfg).to(dev)
    enc = model.geometry.encoding.encoding.encoding
    gen = torch.Generator().manual_seed(0)
    with torch.no_grad():
        enc.params.copy_(((torch.rand(enc.params.numel(), generator=gen) * 2 - 1) * 3e-2).to(dev))
        l0 = model.geometry.network.layers[0]
        l0.weight_v[:, 3:] = (torch.randn(l0.weight_v[:, 3:].shape, generator=gen) * 0.3).to(dev)
    model.train()
    model.geometry.update_step(0, 0)
    model.cos_anneal_ratio = 1.0
    assert model._fused_ok()
    rays = camera_rays(48, 48, seed=11)
    u = torch.rand(rays.shape[0], generator=torch.Generator().manual_seed(12))
    out = model.forward_(rays.to(dev), stratified_u=u.to(dev))
    roi = torch.tensor([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5])
    ri, ts, te = oracle.ray_marching(rays[:, :3].contiguous(), rays[:, 3:].contiguous(), scene_aabb=roi, near_plane=0.0,
                                     far_plane=1e10, render_step_size=model.render_step_size, stratified_u=u)
    assert torch.equal(out["ray_indices"].cpu(), ri) and ri.numel() > 200000
    from test_gpu_model import oracle_params
    meta, table, mlp, var = oracle_params(model)
    eps = model.geometry._finite_difference_eps
    assert abs(eps - 3.0 / 8192 * 1.0) < 1e-3 and eps < 4e-4            # one cell of the finest level
    ref = oracle.neus_geometry_render(rays, ri, ts, te, table, meta, mlp, var, radius=1.5, fd_eps=eps)
    for k in ("opacity", "depth"):
        assert torch.allclose(out[k].cpu(), ref[k], rtol=1e-4, atol=2e-5), k
    assert rel_err(out["sdf_samples"], ref["sdf"]) < 1e-5
    # FD normal at eps = 3.7e-4: one fp32 ulp of SDF disagreement is amplified by 1/eps
    assert float((out["sdf_grad_samples"].cpu() - ref["sdf_grad"]).abs().max()) < 2e-2
    g = torch.Generator().manual_seed(13)
    go, gd = torch.randn(ref["opacity"].shape, generator=g), torch.randn(ref["depth"].shape, generator=g)
    ((ref["opacity"] * go).sum() + (ref["depth"] * gd).sum()).backward()
    ((out["opacity"] * go.to(dev)).sum() + (out["depth"] * gd.to(dev)).sum()).backward()
    gt = enc.params.grad.cpu()
    assert torch.nn.functional.cosine_similarity(gt[None], table.grad[None]).item() > 0.9999
    assert float((gt - table.grad).abs().max()) < 2e-2 * float(table.grad.abs().max())
    lin = [m for m in model.geometry.network.layers if isinstance(m, torch.nn.Linear)]
    for m, p in zip(lin, mlp):
        c = torch.nn.functional.cosine_similarity(m.weight_v.grad.cpu().reshape(1, -1), p["v"].grad.reshape(1, -1)).item()
        assert c > 0.9999, c

    # ---- tight pass on the HIP path's own stencil values: at eps = 3.7e-4 the loose gates above are all 1/eps
    # amplification of forward ulps; downstream of the divide the two implementations agree to SURVEY 8(d)'s tolerances
    from test_gpu_model import assert_grads_tight, hip_sdf7
    sdf7 = hip_sdf7(model, rays, ri, ts, te)
    assert rel_err(sdf7, ref["sdf7"]) < 3e-6
    meta2, table2, mlp2, var2 = oracle_params(model)
    ref2 = oracle.neus_geometry_render(rays, ri, ts, te, table2, meta2, mlp2, var2, radius=1.5, fd_eps=eps, sdf7_given=sdf7)
    for k in ("opacity", "depth"):
        assert torch.allclose(out[k].cpu(), ref2[k], rtol=2e-5, atol=3e-6), k
    assert float((out["sdf_grad_samples"].cpu() - ref2["sdf_grad"]).abs().max()) < 1e-4      # (was 2e-2)
    ((ref2["opacity"] * go).sum() + (ref2["depth"] * gd).sum()).backward()
    hip_named, ref_named = {}, {}
    for i, (m, p) in enumerate(zip(lin, mlp2)):
        for name, key in (("weight_v", "v"), ("weight_g", "g"), ("bias", "b")):
            hip_named[f"{i}.{name}"], ref_named[f"{i}.{name}"] = getattr(m, name).grad, p[key].grad
    hip_named["variance"], ref_named["variance"] = model.variance.variance.grad.reshape(1), var2.grad.reshape(1)
    assert_grads_tight(hip_named, ref_named, gt, table2.grad)


def _stencil_inputs(S, dev, seed=0):
    """Ray-like stencil points at the bench's eps (one finest cell): consecutive samples advance by one marching step."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    n_rays = max(S // 590, 1)
    o = torch.rand(n_rays, 3, generator=g) * 0.2
    d = torch.nn.functional.normalize(torch.rand(n_rays, 3, generator=g) + 0.2, dim=-1)
    step = 0.00507421875 / 3.0
    k = torch.arange(S) % 590
    ray = (torch.arange(S) // 590).clamp(max=n_rays - 1)
    centre = (o[ray] + d[ray] * (k[:, None] * step)).clamp(0, 1)
    eps_unit = 1.0 / 8192 * (8192 / 8173.0)
    offs = torch.tensor([[0, 0, 0], [1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]],
                        dtype=torch.float32) * eps_unit
    x7t = (centre[None, :, :] + offs[:, None, :]).clamp(0.0, 1.0).contiguous()       # [7,S,3] tap-major
    return x7t.to(dev), eps_unit


def _bwd_fd7(x7t, dplanes, meta, n_params, eps_unit):
    from rise_sdf_amd._lib import check, lib, ptr, stream_ptr
    S = x7t.shape[1]
    dt = torch.zeros(n_params, dtype=torch.float32, device=x7t.device)
    nbytes = int(lib().rsdf_hashgrid_bwd_fd7_scratch_bytes(ctypes.byref(meta), S, 16, eps_unit))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=x7t.device)
    check(lib().rsdf_hashgrid_bwd_fd7(ptr(x7t), ptr(dplanes), ctypes.byref(meta), S, 16, eps_unit, ptr(dt), ptr(scratch),
                                      nbytes, stream_ptr()), "hashgrid_bwd_fd7")
    return dt


def test_hash_backward_full_chunk_vs_generic_atomics(dev):
    """One bench chunk: 18.9 M samples x 7 taps x 16 levels.  The stencil path (merge, bin, LDS reduce) must produce
    the table gradient of the generic per-corner atomic kernel on the same points and gradients (fp32 sums in different
    orders: 1e-4 of the largest row), and conserve the checksum sum(dtable) = sum over taps/levels of sum(dplanes)
    (every trilinear weight set sums to 1)."""
    from rise_sdf_amd import _lib
    from rise_sdf_amd._lib import check, lib, ptr, stream_ptr
    S = 18_900_000
    meta, n_params = _lib.make_grid_meta(16, 2, 19, 32, 1.447269237440378)
    x7t, eps_unit = _stencil_inputs(S, dev)
    g = torch.Generator(device=dev).manual_seed(1)
    dplanes = torch.randn(16, 7, S, 2, generator=g, device=dev)
    dt = _bwd_fd7(x7t, dplanes, meta, n_params, eps_unit)
    # generic kernel: rows = 7 S points, dy [7S, 32] with column 2 l + f <- dplanes[l, t, s, f]
    x = x7t.reshape(-1, 3).contiguous()
    dy = dplanes.permute(1, 2, 0, 3).reshape(7 * S, 32).contiguous()
    dt_ref = torch.zeros(n_params, dtype=torch.float32, device=dev)
    check(lib().rsdf_hashgrid_bwd(ptr(x), ptr(dy), ctypes.byref(meta), 7 * S, 16, 32, 0, ptr(dt_ref), stream_ptr()),
          "hashgrid_bwd")
    torch.cuda.synchronize()
    scale = float(dt_ref.abs().max())
    assert float((dt - dt_ref).abs().max()) < 1e-4 * scale, (float((dt - dt_ref).abs().max()), scale)
    total = float(dplanes.double().sum())
    assert abs(float(dt.double().sum()) - total) < 1e-6 * float(dplanes.double().abs().sum())
    del dy, dt_ref


QUEUE_PROBE = """
import ctypes, os, sys, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import oracle
from rise_sdf_amd import _lib
from test_gpu_regimes import _bwd_fd7, _stencil_inputs
dev = torch.device("cuda:0")
S = 40000
cfg = dict(n_levels=16, n_features=2, log2_hashmap_size=19, base_resolution=32, per_level_scale=1.447269237440378)
meta_o, n_params = oracle.grid_meta(**cfg)
meta_g, _ = _lib.make_grid_meta(16, 2, 19, 32, 1.447269237440378)
x7t, eps_unit = _stencil_inputs(S, dev, seed=3)
g = torch.Generator().manual_seed(4)
dplanes = torch.randn(16, 7, S, 2, generator=g)
dt = _bwd_fd7(x7t, dplanes.to(dev), meta_g, n_params, eps_unit).cpu()
x = x7t.cpu().reshape(-1, 3)
dy = dplanes.permute(1, 2, 0, 3).reshape(7 * S, 32)
t = torch.zeros(n_params, requires_grad=True)
(oracle.hashgrid_encode(x, t, meta_o) * dy).sum().backward()
err, scale = float((dt - t.grad).abs().max()), float(t.grad.abs().max())
print("RESULT", err, scale, int((dt != 0).sum()), int((t.grad != 0).sum()))
"""


@pytest.mark.parametrize("qscale", ["0.02", "0.5"])
def test_hash_backward_queue_overflow_path(dev, qscale, tmp_path):
    """RSDF_FD7_QUEUE_SCALE shrinks the per-bin queues to a fraction of the expected record count, so most (0.02) or
    about half (0.5) of the records find their queue full: they must reach the table through the direct-atomic
    fallback, and the reducer must clamp its count to the capacity.  Result vs the fp64 oracle (run in a subprocess: the
    knob is read when the plan is made, and must not leak into other tests)."""
    script = tmp_path / "probe.py"
    script.write_text(QUEUE_PROBE.format(root=ROOT))
    env = dict(os.environ, RSDF_FD7_QUEUE_SCALE=qscale)
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][-1].split()
    err, scale, nz_g, nz_o = float(line[1]), float(line[2]), int(line[3]), int(line[4])
    assert err < 2e-5 * scale + 1e-7, (err, scale)
    assert abs(nz_g - nz_o) <= max(4, int(2e-5 * nz_o))     # (block-float queue records: test_gpu_ops._same_zero_pattern)


def test_bench_two_ranks_on_one_gpu(dev):
    """bench.py's N > 1 path on real hardware.  No multi-GPU node has been available, so two ranks share this GPU
    (RSDF_DIST_SHARE_GPU=1: gloo collectives, RCCL refuses two ranks per device): the self-launcher, the barriers, the
    max/sum reductions of time and samples and the gradient all-reduce on device tensors all execute; the line must
    aggregate both ranks.  Not a performance number."""
    import json
    env = dict(os.environ, RSDF_DIST_SHARE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    common = ["--steps", "1", "--warmup", "1", "--width", "96", "--height", "96", "--chunk", "4608", "--cpu-rays", "0", "--no-extras"]
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common, env=env,
                        capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, r2.stderr[-2000:]
    two = json.loads(r2.stdout.strip().splitlines()[-1])
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common, env=env,
                        capture_output=True, text=True, timeout=900)
    assert r1.returncode == 0, r1.stderr[-2000:]
    one = json.loads(r1.stdout.strip().splitlines()[-1])
    assert two["n_gpus"] == 2 and two["config"]["rccl_ranks"] == 2 and two["config"]["dist_backend"] == "gloo"
    assert two["scaling"] == "weak" and one["n_gpus"] == 1
    # each rank renders its own view of the same box: the job's samples are about twice one rank's
    ratio = two["config"]["samples_per_step"] / one["config"]["samples_per_step"]
    assert 1.6 < ratio < 2.4, ratio


def test_bench_c3_two_ranks_on_one_gpu(dev):
    """``bench.py --workload c3 --gpus 2`` itself (only TrainStep was driven so far): the config[3] training
    step on two ranks sharing this GPU (gloo, RSDF_DIST_SHARE_GPU=1), gradient all-reduce issued asynchronously after the
    backward and finished before Adam; the printed line must aggregate both ranks (samples summed, time max-reduced)."""
    import json
    env = dict(os.environ, RSDF_DIST_SHARE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    lines = {}
    for n in (2, 1):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "c3", "--gpus", str(n), "--steps", "3",
                            "--warmup", "4", "--hidden", "64", "--no-extras", "--cpu-rays", "0"], env=env, capture_output=True,
                           text=True, timeout=1500)
        assert r.returncode == 0, r.stderr[-3000:]
        lines[n] = json.loads(r.stdout.strip().splitlines()[-1])
    two, one = lines[2], lines[1]
    assert two["n_gpus"] == 2 and two["config"]["rccl_ranks"] == 2 and two["config"]["dist_backend"] == "gloo"
    assert two["scaling"] == "weak" and two["steps"] == 3 and two["value"] > 0
    assert one["n_gpus"] == 1 and one["config"]["rccl_ranks"] == 1
    # every rank steers its own batch to the same sample target: the job's samples and rays are about twice one rank's
    for k in ("samples_per_step", "rays_per_step"):
        ratio = two["config"][k] / one["config"][k]
        assert 1.5 < ratio < 2.6, (k, ratio)
    assert abs(two["value"] - two["config"]["samples_per_step"] / (two["ms_per_step"] / 1e3)) < 1e-3 * two["value"]


@pytest.mark.parametrize("which", ["coop", "legacy"])
def test_alternative_mlp_backward_kernels_stay_correct(dev, which):
    """H = 64 ships the quad backward (mlp_quad.hip); the cooperative (mlp_coop.hip, also the H = 32 / 128 kernel) and the
    round-1 per-wave kernels remain selectable with RSDF_MLP_BWD for A/B timing.  They must keep passing the same
    parity tests (the variable is read at launch, so the run is a subprocess)."""
    env = dict(os.environ, RSDF_MLP_BWD=which)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_model.py"), "-q", "-m", "gpu",
                        "-k", "fused_field_with_feature_gradients or neus_render_matches_oracle", "-x"],
                       env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:]


@pytest.mark.parametrize("hidden", [32, 64, 128])
@pytest.mark.parametrize("n_samples,active,frozen", [(5, 6, False), (33, 3, False), (1000, 6, True)])
def test_fused_kernels_edge_shapes(dev, hidden, n_samples, active, frozen):
    """Tile remainders (5 and 33 samples: fewer rows than a tile / one row over), masked levels (3 of 6 active) and a
    frozen table (no d_planes requested): the fused stencil kernels against the per-layer path on the same weights
    (itself checked against the oracle in test_gpu_model.py)."""
    import rise_sdf_amd as R
    from test_gpu_model import model_config
    torch.manual_seed(7)
    cfg = model_config(hidden=hidden, n_levels=6, feat=13)
    geo = R.make("volume-sdf", cfg.geometry).to(dev)
    geo.train()
    with torch.no_grad():
        geo.encoding.encoding.encoding.params.mul_(300.0)
        l0 = geo.network.layers[0]
        l0.weight_v[:, 3:] = torch.randn_like(l0.weight_v[:, 3:]) * 0.3
    geo.update_step(0, 0)
    if active < 6:
        geo.encoding.encoding.current_level = active            # ProgressiveBandHashGrid: levels >= active read as zero
        geo.encoding.encoding.mask = torch.cat([torch.ones(2 * active), torch.zeros(2 * (6 - active))])
    enc = geo.encoding.encoding.encoding
    enc.params.requires_grad_(not frozen)
    g = torch.Generator().manual_seed(n_samples)
    ro = (torch.rand(1, 3, generator=g) - 0.5).to(dev)
    rd = torch.nn.functional.normalize(torch.randn(1, 3, generator=g), dim=-1).to(dev)
    ts = (torch.sort(torch.rand(n_samples, generator=g))[0] * 1.2).to(dev)
    te = ts + 0.004
    ri = torch.zeros(n_samples, dtype=torch.int64, device=dev)
    gs = torch.randn(7, n_samples, generator=g).to(dev)
    gf = torch.randn(n_samples, 13, generator=g).to(dev)

    def grads():
        out = {n: (p.grad.clone() if p.grad is not None else None) for n, p in geo.named_parameters()}
        for p in geo.parameters():
            p.grad = None
        return out

    sdf7t, feat = geo.sdf7_from_rays(ro, rd, ri, ts, te, want_feature=True)
    ((sdf7t * gs).sum() + (feat * gf).sum()).backward()
    g_fused = grads()
    out7 = geo.field7_from_rays(ro, rd, ri, ts, te)                      # [7S, 13], rows 7 i + t
    ref7 = out7.view(n_samples, 7, 13)
    ((ref7[:, :, 0].t() * gs).sum() + (ref7[:, 0, :] * gf).sum()).backward()
    g_ref = grads()
    assert torch.allclose(sdf7t, ref7[:, :, 0].t(), rtol=1e-5, atol=1e-6)
    assert torch.allclose(feat, ref7[:, 0, :], rtol=1e-5, atol=1e-6)
    for name in g_ref:
        if g_ref[name] is None:
            assert g_fused[name] is None, name
            continue
        scale = float(g_ref[name].abs().max()) + 1e-12
        assert float((g_fused[name] - g_ref[name]).abs().max()) < 2e-4 * scale + 1e-7, (name, hidden)


def test_two_stream_step_accumulates_the_same_gradients(dev):
    """bench.py's run_step with the chunks alternating over two HIP streams (the shipped default) against one chunk at a
    time: same sample total, same accumulated gradients (float-atomic order only), for every parameter."""
    import rise_sdf_amd as R
    sys.path.insert(0, ROOT)
    import bench
    torch.manual_seed(0)
    cfg = bench.c1_config(hidden=64)
    cfg["num_samples_per_ray"] = 256
    model = R.make("neus", cfg).to(dev)
    enc = model.geometry.encoding.encoding.encoding
    gen = torch.Generator().manual_seed(0)
    with torch.no_grad():
        enc.params.copy_(((torch.rand(enc.params.numel(), generator=gen) * 2 - 1) * 3e-2).to(dev))
        l0 = model.geometry.network.layers[0]
        l0.weight_v[:, 3:] = (torch.randn(l0.weight_v[:, 3:].shape, generator=gen) * 0.3).to(dev)
    model.train()
    model.geometry.update_step(0, 0)
    rays = camera_rays(64, 64, seed=5).to(dev)
    n = rays.shape[0]
    u = torch.rand(n, generator=gen).to(dev)
    cot = [torch.randn(n, 1, generator=gen).to(dev), torch.randn(n, 1, generator=gen).to(dev),
           torch.randn(n, 3, generator=gen).to(dev)]
    res = []
    for streams in (1, 2, 3):
        for p in model.parameters():
            p.grad = None
        total = bench.run_step(model, rays, u, cot, 512, streams=streams)
        torch.cuda.synchronize()
        res.append((total, {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
    assert res[0][0] == res[1][0] == res[2][0] > 100000
    for total, grads in res[1:]:
        assert set(grads) == set(res[0][1])
        for k, g in grads.items():
            ref = res[0][1][k]
            assert float((g - ref).abs().max()) < 2e-5 * float(ref.abs().max()) + 1e-12, k
