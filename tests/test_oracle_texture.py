"""CPU: the oracle's radiance-branch restatements vs golden vectors from the imported reference."""
import os

import numpy as np
import torch

from oracle import texture as otex


def load(golden_dir, name):
    return {k: torch.tensor(v) for k, v in np.load(os.path.join(golden_dir, name)).items()}


def nets_from(z, prefix="p__"):
    def net(name, idxs):
        return [{"w": z[f"{prefix}{name}_network_layers_{i}_weight"], "b": z[f"{prefix}{name}_network_layers_{i}_bias"]}
                for i in idxs]
    return {"albedo": net("albedo", (0, 2, 4, 6, 8)), "metallic": net("metallic", (0, 2, 4)),
            "roughness": net("roughness", (0, 2, 4)), "env": net("env", (0, 2, 4, 6, 8))}


def test_frequency_and_srgb_match_reference(golden_dir):
    z = load(golden_dir, "freq_srgb.npz")
    assert torch.allclose(otex.vanilla_frequency(z["x"], 6), z["freq"], rtol=0, atol=1e-6)
    assert torch.allclose(otex.rgb_to_srgb(z["lin"]), z["srgb"], rtol=0, atol=1e-7)


def test_texture_stage0_matches_reference(golden_dir):
    """VolumeMixedMipSplitOcc.forward(stage=0) of the reference (SH stand-in = the oracle's basis)."""
    z = load(golden_dir, "texture_stage0.npz")
    nets = nets_from(z)
    for n in nets.values():
        for p in n:
            p["w"].requires_grad_(True), p["b"].requires_grad_(True)
    feats, nrm = z["features"].clone().requires_grad_(True), z["normals"].clone().requires_grad_(True)
    col = otex.texture_stage0(feats, z["dirs"], nrm, z["positions"], nets)
    assert torch.allclose(col, z["colors"], rtol=1e-5, atol=1e-6)
    (col * z["gcolors"]).sum().backward()
    assert torch.allclose(feats.grad, z["g_features"], rtol=1e-4, atol=1e-6)
    assert torch.allclose(nrm.grad, z["g_normals"], rtol=1e-4, atol=1e-6)
    ref = z["g__albedo_network_layers_0_weight"]
    assert torch.allclose(nets["albedo"][0]["w"].grad, ref, rtol=1e-4, atol=1e-6 * float(ref.abs().max() + 1))
    assert float(z["g__roughness_network_layers_0_weight"].abs().max()) == 0.0  # unused at stage 0


def test_sh_basis_is_orthonormal():
    """The (unpinned) SH basis must at least be orthonormal on the sphere: Monte-Carlo Gram matrix ~ I."""
    g = torch.Generator().manual_seed(0)
    d = torch.nn.functional.normalize(torch.randn(400000, 3, generator=g, dtype=torch.float64), dim=-1)
    Y = otex.sh_encode((d + 1) / 2, 5)
    gram = (Y.T @ Y) / d.shape[0] * 4 * np.pi
    assert torch.allclose(gram, torch.eye(25, dtype=torch.float64), atol=2e-2)


def seeded_nets(dtype=torch.float32):
    """The 128-wide radiance networks of the ``*_n128`` fixtures, regenerated from their parameter names
    (tests/helpers.py::seeded_param, as tests/golden/make_golden.py filled the reference module)."""
    from helpers import seeded_param
    shapes = {"albedo": (84, 4, 6), "metallic": (84, 2, 2), "roughness": (84, 2, 1), "env": (73, 4, 3), "secondary": (76, 4, 3)}
    nets = {}
    for name, (k, nh, nout) in shapes.items():
        dims = [k] + [128] * nh + [nout]
        nets[name] = [{"w": seeded_param(f"{name}_network.layers.{2 * i}.weight", (dims[i + 1], dims[i]), 128).to(dtype),
                       "b": seeded_param(f"{name}_network.layers.{2 * i}.bias", (dims[i + 1],), 128).to(dtype)}
                      for i in range(nh + 1)]
    return nets


def seeded_base():
    from helpers import seeded_param
    return seeded_param("emitter.base", (6, 64, 64, 3), seed=128).abs() * 8.0 + 0.05


def test_texture_stage0_n128_matches_reference(golden_dir):
    """The reference's VolumeMixedMipSplitOcc.forward(stage=0) at the widths of the shipped yaml (128-wide networks, 48
    features) with a cotangent spanning eight decades: outputs and EVERY parameter gradient."""
    z = load(golden_dir, "texture_stage0_n128.npz")
    nets = seeded_nets()
    for n in nets.values():
        for p in n:
            p["w"].requires_grad_(True), p["b"].requires_grad_(True)
    feats, nrm = z["features"].clone().requires_grad_(True), z["normals"].clone().requires_grad_(True)
    col = otex.texture_stage0(feats, z["dirs"], nrm, z["positions"], nets)
    assert torch.allclose(col, z["colors"], rtol=1e-5, atol=1e-6)
    (col * z["gcolors"]).sum().backward()
    assert torch.allclose(feats.grad, z["g_features"], rtol=1e-4, atol=1e-6 * float(z["g_features"].abs().max()))
    for name in ("albedo", "metallic", "env"):
        for i, p in enumerate(nets[name]):
            for key, suffix in (("w", "weight"), ("b", "bias")):
                ref = z[f"g__{name}_network_layers_{2 * i}_{suffix}"]
                assert float((p[key].grad - ref).abs().max()) <= 1e-5 * float(ref.abs().max()), (name, i, key)
