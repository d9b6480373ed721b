"""The training step's host reads.  Capacity-mode sampling (one read per sampling call instead of
two) and the masked secondary-ray blend (no ``torch.nonzero``) must produce exactly what the exact paths produce."""
import pytest
import torch

from helpers import camera_rays, sphere_binary
from test_gpu_model import split_config

pytestmark = pytest.mark.gpu


def _model(dev, indirect=True, stage1=False):
    import rise_sdf_amd as R
    torch.manual_seed(0)
    cfg = split_config(indirect=indirect)
    cfg["curvature"] = False
    if stage1:
        cfg["split_sum_kick_in_step"] = 0
        cfg["light"] = {"name": "envlight-mip-cube", "envlight_config": {"scale": 0.5, "bias": 0.25, "base_res": 64,
                                                                           "hdr_filepath": None}}
    model = R.make("split-mixed-occ", cfg).to(dev)
    model.train()
    with torch.no_grad():
        model.geometry.encoding.encoding.encoding.params.mul_(1000.0)
        l0 = model.geometry.network.layers[0]
        l0.weight_v[:, 3:] = torch.randn_like(l0.weight_v[:, 3:]) * 0.3
        model.variance.variance.fill_(0.6)
    model.occupancy_grid.binaries = sphere_binary(128, 0.2, 0.9).to(dev)[None]
    model.background_color = torch.ones(3, device=dev)
    model.update_step(0, 0)
    return model


def test_capacity_mode_sampling_is_identical(dev):
    model = _model(dev)
    grid = model.occupancy_grid
    rays = camera_rays(24, 24, seed=2).to(dev)
    ro, rd = rays[:, :3].contiguous(), rays[:, 3:].contiguous()
    u = torch.rand(rays.shape[0], generator=torch.Generator().manual_seed(3)).to(dev)
    kw = dict(alpha_fn=model._alpha_fn(ro, rd), render_step_size=model.render_step_size, stratified_u=u, cone_angle=0.0,
              alpha_thre=0.0)
    with torch.no_grad():
        exact = grid.sampling(ro, rd, **kw)
        grid.capacity_mode = True
        first = grid.sampling(ro, rd, **kw)                  # no capacity yet: exact path, remembers the size
        assert grid.stats["capped_calls"] == 0
        capped = grid.sampling(ro, rd, **kw)                 # sized from the previous call: one host read
        assert grid.stats["capped_calls"] == 1 and grid.stats["overflows"] == 0
        for k in list(grid._capacity):                       # a capacity that is too small: detected, redone exactly
            grid._capacity[k] = 1000
        over = grid.sampling(ro, rd, **kw)
        assert grid.stats["overflows"] == 1
        again = grid.sampling(ro, rd, **kw)                  # ... and the capacity has grown back
        assert grid.stats["overflows"] == 1 and grid.stats["capped_calls"] == 3
    assert exact[0].numel() > 3000
    for other in (first, capped, over, again):
        for a, b in zip(exact, other):
            assert torch.equal(a, b)


@pytest.mark.parametrize("stage1", [False, True])
def test_masked_secondary_blend_matches_the_gather_path(dev, stage1):
    rays = camera_rays(20, 20, seed=2).to(dev)
    u = torch.rand(rays.shape[0], generator=torch.Generator().manual_seed(3)).to(dev)
    g = torch.randn(rays.shape[0], 3, generator=torch.Generator().manual_seed(4)).to(dev)
    key = "comp_rgb_phys_full" if stage1 else "comp_rgb_full"
    res = []
    for masked in (False, True):
        model = _model(dev, stage1=stage1)
        model.masked_secondary = masked
        model.occupancy_grid.capacity_mode = masked
        for _ in range(3 if masked else 1):                  # later passes: the capped / read-free samplers are in use
            for p in model.parameters():
                p.grad = None
            if stage1:
                model.emitter.build_mips()
            out = model.forward_(rays, stratified_u=u)
            (out[key] * g).sum().backward()
        if masked:
            st = model.occupancy_grid.stats
            assert st["capped_calls"] >= 1 and st["blind_calls"] >= 1    # primary: one read; secondary: none
            assert st["overflows"] == 0 and st["blind_overflows"] == 0
            assert int(model._last_secondary["valid"].sum()) > 40
        res.append((out, {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}))
    (o0, g0), (o1, g1) = res
    for k in ("comp_rgb", "comp_spec_rgb", "comp_rgb_full", "opacity", "depth") + (("comp_rgb_phys", key) if stage1 else ()):
        assert torch.allclose(o0[k], o1[k], rtol=0, atol=1e-6), k
    assert set(g0) == set(g1)
    for n in g0:
        scale = float(g0[n].abs().max()) + 1e-20
        assert float((g0[n] - g1[n]).abs().max()) < 2e-5 * scale, n     # float atomics: last bits only


@pytest.mark.parametrize("masked", [False, True])
def test_secondary_rays_reuse_the_samplers_alphas_bit_for_bit(dev, monkeypatch, masked):
    """models/volrend.py:60-75 evaluates the field a second time for the secondary samples the visibility test has just kept;
    here secondary_rendering takes the sampler's own alphas (compacted alongside the samples): the occlusion maps and every
    output are bit-identical to the recomputing form (RSDF_SECONDARY_REUSE_ALPHA=0), on the exact, the capped and the read-free
    sampling paths."""
    rays = camera_rays(20, 20, seed=2).to(dev)
    u = torch.rand(rays.shape[0], generator=torch.Generator().manual_seed(3)).to(dev)
    res = []
    for reuse in ("0", "1"):
        monkeypatch.setenv("RSDF_SECONDARY_REUSE_ALPHA", reuse)
        model = _model(dev, stage1=False)
        model.masked_secondary = masked
        model.occupancy_grid.capacity_mode = masked
        outs = []
        for _ in range(3 if masked else 1):
            with torch.no_grad():
                out = model.forward_(rays, stratified_u=u)
            outs.append({k: out[k].clone() for k in ("comp_rgb", "comp_spec_rgb", "comp_rgb_full", "opacity")})
            outs[-1]["tr"] = model._last_secondary["tr"].clone()
            outs[-1]["sec_depth"] = model._last_secondary["sec_depth"].clone()
        if masked:
            assert model.occupancy_grid.stats["blind_calls"] >= 1
        res.append(outs)
    assert float(res[0][-1]["tr"].min()) < 0.5                     # some secondary rays ARE occluded
    for a, b in zip(*res):
        for k in a:
            assert torch.equal(a[k], b[k]), k


def test_blind_overflow_warns_and_recovers_on_the_exact_path(dev):
    """a read-free secondary pass whose candidates outgrow the remembered capacity is counted, warned about, and
    the NEXT pass takes the exact path (and re-measures); a larger ray batch scales the capacity instead of overflowing."""
    import warnings
    model = _model(dev, stage1=False)
    model.masked_secondary = True
    est = model.occupancy_grid
    est.capacity_mode = True
    rays = camera_rays(20, 20, seed=2).to(dev)
    u = torch.rand(rays.shape[0], generator=torch.Generator().manual_seed(3)).to(dev)
    out0 = model.forward_(rays, stratified_u=u)                      # exact first pass: capacities measured
    model.forward_(rays, stratified_u=u)                             # blind pass in use
    assert est.stats["blind_calls"] >= 1 and est.stats["blind_overflows"] == 0
    ref_tr = model._last_secondary["tr"].clone()
    bkeys = [k for k in est._capacity if k[0] == "blind"]
    assert bkeys
    for k in bkeys:                                                   # force an overflow
        est._capacity[k] = 64
    est._pending = []                                                 # (the previous pass's counts would re-measure the capacity)
    model.forward_(rays, stratified_u=u)                             # truncated pass (cannot be redone: per-ray results used)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        model.forward_(rays, stratified_u=u)                         # its counts arrive with this step's read -> warning; this
        assert any("outgrew its buffers" in str(x.message) for x in w)  #   pass itself already runs exactly
    assert est.stats["blind_overflows"] >= 1
    assert torch.allclose(model._last_secondary["tr"], ref_tr, atol=1e-6)
    n_blind = est.stats["blind_calls"]
    model.forward_(rays, stratified_u=u)                             # capacity re-measured: blind again, no overflow
    assert est.stats["blind_calls"] == n_blind + 1
    # four times the rays: the capacity is scaled by the ray-count ratio instead of overflowing
    rays4 = camera_rays(40, 40, seed=2).to(dev)
    u4 = torch.rand(rays4.shape[0], generator=torch.Generator().manual_seed(3)).to(dev)
    n_over = est.stats["blind_overflows"]
    model.forward_(rays4, stratified_u=u4)
    model.forward_(rays4, stratified_u=u4)
    assert est.stats["blind_overflows"] == n_over
    assert out0["comp_rgb"].shape[0] == rays.shape[0]


def test_eval_render_on_two_streams_is_identical(dev):
    """model.eval()(rays): chunk_batch alternates the ray chunks over two HIP streams (config ``eval_streams``, default 2);
    the rendered maps equal the one-stream render bit for bit."""
    import time
    model = _model(dev, indirect=True, stage1=True)
    model.eval()
    model.config["ray_chunk"] = 512
    rays = camera_rays(64, 64, seed=7).to(dev)
    outs, times = [], []
    with torch.no_grad():
        model.emitter.build_mips()
        for streams in (1, 2, 1, 2):
            model.config["eval_streams"] = streams
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            outs.append(model(rays))
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
    for k in ("comp_rgb_full", "comp_rgb_phys_full", "opacity", "depth", "comp_normal"):
        assert torch.equal(outs[2][k], outs[3][k]), k
        assert torch.equal(outs[0][k], outs[3][k]), k
    print("eval render of 4096 rays in 512-ray chunks: one stream %.1f ms, two streams %.1f ms" % (times[2] * 1e3, times[3] * 1e3))


def test_training_step_with_the_prefilter_on_a_side_stream(dev):
    """TrainStep issues build_mips behind the sampling read, on a side stream (the prefilter's vector-ALU work beside the
    networks' matrix / memory work; its backward runs on that stream too).  With and without the side stream from the same
    seeds: the first step's loss (a pure forward) is bitwise the same and so are the sample counts; the second step's loss
    -- after one backward + Adam update -- agrees to what the order of the float atomics in the gradient kernels leaves
    open between any two runs (measured: 6e-7 relative; Adam with eps 1e-12 amplifies it from the third step on)."""
    from rise_sdf_amd.step import build_synthetic_training

    def run(side):
        model, ts = build_synthetic_training(dev, stage=1, hidden=64, views=2, res=48, seed=3, grad_buckets=False,
                                             model_overrides={"light": {"name": "envlight-mip-cube", "envlight_config": {
                                                 "hdr_filepath": None, "clamp": True, "nmf_format": False, "scale": 0.5,
                                                 "bias": 0.25, "base_res": 64}}})
        if not side:
            ts.prefilter_stream = None
        assert (ts.prefilter_stream is not None) == side
        out = []
        for k in range(2):
            r = ts.step(20000 + k)
            out.append((float(r["loss"]), r["num_samples"]))
        g = model.emitter.base.grad.detach().clone()
        torch.cuda.synchronize()
        return out, g

    a, ga = run(True)
    b, gb = run(False)
    c, gc = run(False)
    assert a[0] == b[0] == c[0], (a, b, c)
    assert a[1][1] == b[1][1] == c[1][1]
    noise = abs(b[1][0] - c[1][0]) / abs(b[1][0])
    assert abs(a[1][0] - b[1][0]) / abs(b[1][0]) < max(1e-5, 10 * noise), (a, b, c)
    scale = float(gb.abs().max())
    assert float((ga - gb).abs().max()) <= max(1e-4 * scale, 10 * float((gb - gc).abs().max())), "d loss / d light"
