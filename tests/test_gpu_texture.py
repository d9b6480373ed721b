"""GPU parity for the radiance-branch kernels (csrc/texture.hip) and the VolumeMixedMipSplitOcc mirror."""
import os

import numpy as np
import pytest
import torch

from oracle import texture as otex
from helpers import rel_err

pytestmark = pytest.mark.gpu


def test_freq_sh_reflect_srgb_kernels(dev):
    from rise_sdf_amd import texture_ops as T
    g = torch.Generator().manual_seed(0)
    x = (torch.rand(3000, 3, generator=g) * 2 - 1) * 1.5
    # fp32 sin/cos of arguments up to 2^5 * 1.5 = 48 rad: 1e-5 absolute
    assert torch.allclose(T.freq_encode(x.to(dev), 6).cpu(), otex.vanilla_frequency(x, 6), rtol=0, atol=1e-5)
    mask = torch.tensor([1.0, 1.0, 0.7, 0.2, 0.0, 0.0])
    assert torch.allclose(T.freq_encode(x.to(dev), 6, mask=mask).cpu(), otex.vanilla_frequency(x, 6, mask=mask),
                          rtol=0, atol=1e-5)
    lin = torch.rand(5000, 3, generator=g) * 1.3
    lo, lg = lin.clone().requires_grad_(True), lin.to(dev).requires_grad_(True)
    gy = torch.randn(5000, 3, generator=g)
    (otex.rgb_to_srgb(lo) * gy).sum().backward()
    yg = T.rgb_to_srgb(lg)
    (yg * gy.to(dev)).sum().backward()
    assert torch.allclose(yg.cpu(), otex.rgb_to_srgb(lin), rtol=1e-5, atol=1e-6)
    assert torch.allclose(lg.grad.cpu(), lo.grad, rtol=1e-4, atol=1e-5)

    d = torch.nn.functional.normalize(torch.randn(4000, 3, generator=g), dim=-1)
    n = torch.nn.functional.normalize(torch.randn(4000, 3, generator=g), dim=-1)
    for degree in (4, 5):
        no, ng = n.clone().requires_grad_(True), n.to(dev).requires_grad_(True)
        wo_o, nov_o = otex.reflect_dirs(d, no)
        sh_o = otex.sh_encode((wo_o + 1) / 2, degree)
        gs, gn = torch.randn(sh_o.shape, generator=g), torch.randn(4000, 1, generator=g)
        ((sh_o * gs).sum() + (nov_o * gn).sum()).backward()
        wo01, nov_g = T.reflect(d.to(dev), ng)
        sh_g = T.sh_encode(wo01, degree)
        ((sh_g * gs.to(dev)).sum() + (nov_g * gn.to(dev)).sum()).backward()
        assert torch.allclose(sh_g.cpu(), sh_o, rtol=1e-5, atol=1e-5)
        assert torch.allclose(nov_g.cpu(), nov_o, rtol=1e-6, atol=1e-6)
        assert rel_err(ng.grad, no.grad) < 1e-5


def test_texture_stage0_reference_fixture(dev, golden_dir):
    """Reference VolumeMixedMipSplitOcc.forward(stage=0) (tests/golden/texture_stage0.npz) vs the HIP mirror
    with the same weights: colours and every gradient."""
    import rise_sdf_amd as R
    z = {k: torch.tensor(v) for k, v in np.load(os.path.join(golden_dir, "texture_stage0.npz")).items()}
    mlp = lambda n: {"otype": "VanillaMLP", "activation": "ReLU", "output_activation": "none", "n_neurons": 64,
                     "n_hidden_layers": n}
    cfg = R.Config({
        "name": "volume-mixed-mip-split-occ", "input_feature_dim": 13, "other_dim": 3, "sample_size": 8,
        "dir_encoding_config": {"otype": "SphericalHarmonics", "degree": 5, "reflected": True},
        "metallic_mlp_network_config": mlp(2), "albedo_mlp_network_config": mlp(4),
        "spec_mlp_network_config": mlp(4), "roughness_mlp_network_config": mlp(2),
        "secondary_mlp_network_config": mlp(4),
        "xyz_encoding_config": {"otype": "VanillaFrequency", "n_frequencies": 6}, "color_activation": "sigmoid"})
    tex = R.make("volume-mixed-mip-split-occ", cfg).to(dev)
    sd = {k[3:]: v for k, v in z.items() if k.startswith("p__")}
    with torch.no_grad():
        for name, p in tex.named_parameters():
            p.copy_(sd[name.replace(".", "_")])
    feats = z["features"].to(dev).requires_grad_(True)
    nrm = z["normals"].to(dev).requires_grad_(True)
    col = tex(feats, z["dirs"].to(dev), nrm, z["positions"].to(dev), None, 0)
    # fp32 radiance within 1e-4 relative (north_star)
    assert torch.allclose(col.cpu(), z["colors"], rtol=1e-4, atol=1e-6)
    (col * z["gcolors"].to(dev)).sum().backward()
    assert rel_err(feats.grad, z["g_features"]) < 1e-4
    assert rel_err(nrm.grad, z["g_normals"]) < 1e-4
    for name, p in tex.named_parameters():
        ref = z["g__" + name.replace(".", "_")]
        got = torch.zeros_like(ref) if p.grad is None else p.grad.cpu()
        assert float((got - ref).abs().max()) <= 1e-4 * float(ref.abs().max()) + 1e-7, name


def test_compose_srgb_matches_the_unfused_chain_and_the_oracle(dev):
    """O1 (models/split_mixed_occ.py:405-436): clamp(rgb_to_srgb(comp + bg (1 - opacity)), 0, 1) as one kernel each way ==
    the unfused chain of the same kernels bit for bit, and the oracle's chain within fp32 rounding; values around both
    clamp bounds and the sRGB knee."""
    from rise_sdf_amd import texture_ops as T
    g = torch.Generator().manual_seed(5)
    n = 4097
    comp = torch.rand(n, 3, generator=g) * 1.6 - 0.2           # below 0 and above 1 after composition
    comp[:64] = torch.rand(64, 3, generator=g) * 0.006          # around the 0.0031308 knee
    op = torch.rand(n, 1, generator=g)
    op[:16] = 1.0
    bg = torch.rand(3, generator=g)
    gy = torch.randn(n, 3, generator=g)
    co, oo = comp.clone().requires_grad_(True), op.clone().requires_grad_(True)
    yo = otex.rgb_to_srgb(co + bg[None, :] * (1.0 - oo)).clamp(0, 1)
    (yo * gy).sum().backward()
    cu, ou = comp.to(dev).requires_grad_(True), op.to(dev).requires_grad_(True)
    yu = T.rgb_to_srgb(cu + bg.to(dev)[None, :].expand(n, 3) * (1.0 - ou)).clamp(0, 1)
    (yu * gy.to(dev)).sum().backward()
    cf, of = comp.to(dev).requires_grad_(True), op.to(dev).requires_grad_(True)
    yf = T.compose_srgb(cf, bg.to(dev), of)
    (yf * gy.to(dev)).sum().backward()
    assert torch.equal(yf, yu), "fused compose differs from the unfused chain"
    assert torch.allclose(yf.cpu(), yo.detach(), rtol=1e-5, atol=1e-6)
    assert torch.allclose(cf.grad, cu.grad, rtol=1e-6, atol=1e-7)
    assert torch.allclose(of.grad, ou.grad, rtol=1e-5, atol=1e-6)
    assert torch.allclose(cf.grad.cpu(), co.grad, rtol=1e-4, atol=1e-5)
    assert torch.allclose(of.grad.cpu(), oo.grad, rtol=1e-4, atol=1e-5)


def _seeded_texture_128(dev):
    """The HIP mirror of VolumeMixedMipSplitOcc at the yaml's widths, filled like the reference module of the ``*_n128``
    fixtures (tests/golden/make_golden.py): parameters regenerated from their names."""
    import rise_sdf_amd as R
    from helpers import seeded_param
    from oracle import texture as otex
    mlp = lambda n: {"otype": "VanillaMLP", "activation": "ReLU", "output_activation": "none", "n_neurons": 128,   # noqa: E731
                     "n_hidden_layers": n}
    cfg = R.Config({
        "name": "volume-mixed-mip-split-occ", "input_feature_dim": 48, "other_dim": 3, "sample_size": 8,
        "dir_encoding_config": {"otype": "SphericalHarmonics", "degree": 5, "reflected": True},
        "metallic_mlp_network_config": mlp(2), "albedo_mlp_network_config": mlp(4),
        "spec_mlp_network_config": mlp(4), "roughness_mlp_network_config": mlp(2),
        "secondary_mlp_network_config": mlp(4),
        "xyz_encoding_config": {"otype": "VanillaFrequency", "n_frequencies": 6}, "color_activation": "sigmoid"})
    tex = R.make("volume-mixed-mip-split-occ", cfg).to(dev)
    with torch.no_grad():
        for name, p in tex.named_parameters():
            p.copy_(seeded_param(name, tuple(p.shape), seed=128))
        tex.FG_LUT.copy_(otex.synthetic_fg_lut())
    return tex


def _check_param_grads(tex, z, tol, skip=()):
    worst = ("", 0.0)
    for name, p in tex.named_parameters():
        ref = z["g__" + name.replace(".", "_")]
        if any(s in name for s in skip):
            continue
        got = torch.zeros_like(ref) if p.grad is None else p.grad.cpu()
        err = float((got - ref).abs().max()) / (float(ref.abs().max()) + 1e-30)
        if float(ref.abs().max()) == 0.0:
            assert float(got.abs().max()) == 0.0, name
            continue
        worst = max(worst, (name, err), key=lambda t: t[1])
        assert err <= tol, (name, err)
    return worst


@pytest.mark.parametrize("stage", [0, 1])
def test_pair_kernels_vs_reference_run_gradients_n128(dev, golden_dir, stage):
    """Reference-run GRADIENT fixtures for the layer-pair kernels: the reference's
    VolumeMixedMipSplitOcc.forward at n_neurons 128 / 48 features (tests/golden/texture_stage{0,1}_n128.npz) against the HIP
    mirror, whose five networks must run on rsdf_pair_fwd / rsdf_pair_bwd; the cotangent rows span 1e-8 ... 1 with exact
    zeros (composite weights of a pruned ray), which is what the pair backward's shared gradient-image scale has to survive."""
    import os
    import numpy as np
    import rise_sdf_amd as R
    from rise_sdf_amd import _lib
    from helpers import rel_err, seeded_param
    z0 = {k: torch.tensor(v) for k, v in np.load(os.path.join(golden_dir, "texture_stage0_n128.npz")).items()}
    z = z0 if stage == 0 else {k: torch.tensor(v) for k, v in
                               np.load(os.path.join(golden_dir, "texture_stage1_n128.npz")).items()}
    tex = _seeded_texture_128(dev)
    light = None
    if stage == 1:
        light = R.make("envlight-mip-cube", R.Config(
            {"envlight_config": {"scale": 0.5, "bias": 0.25, "base_res": 64, "hdr_filepath": None}})).to(dev)
        with torch.no_grad():
            light.base.copy_(seeded_param("emitter.base", (6, 64, 64, 3), seed=128).abs() * 8.0 + 0.05)
        light.build_mips()
    feats = z0["features"].to(dev).requires_grad_(True)
    nrm = z0["normals"].to(dev).requires_grad_(True)
    timer = _lib.KernelTimer()
    _lib.set_timer(timer)
    try:
        col = tex(feats, z0["dirs"].to(dev), nrm, z0["positions"].to(dev), light, stage)
        (col * z["gcolors"].to(dev)).sum().backward()
    finally:
        _lib.set_timer(None)
    torch.cuda.synchronize()
    calls = {k: v["calls"] for k, v in timer.summary().items()}
    n_nets = 3 if stage == 0 else 4                      # albedo, metallic, env (+ roughness at stage 1)
    assert calls.get("rsdf_pair_fwd", 0) >= n_nets and calls.get("rsdf_pair_bwd", 0) >= n_nets, calls
    assert torch.allclose(col.cpu(), z["colors"], rtol=1e-4, atol=1e-5), float((col.cpu() - z["colors"]).abs().max())
    assert rel_err(feats.grad, z["g_features"]) < 1e-4
    assert rel_err(nrm.grad, z["g_normals"]) < 1e-3
    if stage == 1:
        assert rel_err(light.base.grad, z["g_base"]) < 1e-4
    # MLP parameter gradients within 1e-4 of the tensor's largest entry
    worst = _check_param_grads(tex, z, 1e-4)
    print("stage %d: worst parameter gradient %s %.2e; entry points %s" % (stage, worst[0], worst[1],
          {k: v for k, v in calls.items() if "pair" in k}))
    R.check_status()


def test_encoders_store_whole_rows_at_any_width(dev):
    """The frequency / SH encoders stage a workgroup's rows in LDS and store along the rows (round 6): every row count around
    the workgroup size, outputs inside a wider matrix (col_off, ld), more than 10 frequencies (more than 64 KB of LDS), and
    the limit of the staging (24 frequencies) as a named error."""
    from rise_sdf_amd import _lib
    from rise_sdf_amd import texture_ops as T
    for n in (1, 255, 256, 257, 1000):
        x = torch.rand(n, 3, generator=torch.Generator().manual_seed(n)) * 2 - 1
        for nf in (1, 6, 12):
            out = torch.full((n, 6 * nf + 5), 7.0, device=dev)
            T.freq_encode(x.to(dev), nf, out=out, col_off=3)
            want = otex.vanilla_frequency(x, nf)
            assert torch.allclose(out[:, 3:3 + 6 * nf].cpu(), want, rtol=0, atol=2e-4 if nf > 8 else 1e-5)   # (2^11 x: fp32 argument)
            assert bool((out[:, :3] == 7.0).all()) and bool((out[:, 3 + 6 * nf:] == 7.0).all())
        d = torch.rand(n, 3, generator=torch.Generator().manual_seed(n + 1))
        for deg in (1, 3, 4, 5):
            assert torch.allclose(T.sh_encode(d.to(dev), deg).cpu(), otex.sh_encode(d, deg), rtol=0, atol=1e-5)
    with pytest.raises(_lib.RiseSdfHipError, match="24 frequencies"):
        T.freq_encode(torch.zeros(4, 3, device=dev), 25)
