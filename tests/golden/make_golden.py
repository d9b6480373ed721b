#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the reference Python in the build container.

Run here only (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_golden.py

The reference's third-party imports that are absent from this image (tinycudann,
pytorch_lightning, omegaconf, nvdiffrast, nerfacc 0.5.3, ...) are replaced by empty
``sys.modules`` stubs so the reference's *own* pure-PyTorch code can run on CPU.  What
is pinned by each fixture, and what is not:

  vanilla_mlp.npz      models/network_utils.py:109-157 VanillaMLP (sphere init + weight_norm +
                       Softplus(100)) -- fully reference code.
  get_alpha.npz        models/split_mixed_occ.py:151-177 get_alpha -- fully reference code.
  volume_sdf_fd.npz    models/geometry.py:206-244 VolumeSDF.forward (contraction, FD taps,
                       clamp, include_xyz, progressive mask) -- reference code, with
                       tcnn.Encoding stubbed by the ORACLE hash grid (tiny-cuda-nn is absent:
                       the encoding values themselves stay unpinned).
  rays.npz             models/ray_utils.py:9-56 -- fully reference code.
  rendering.npz        models/volrend.py:739-895 rendering_with_normals_sdf -- reference
                       orchestration over stubbed nerfacc-0.5.3 calls (index_add / serial product),
                       pins channel order and depth convention only.
  freq_srgb.npz        models/network_utils.py:14-40 VanillaFrequency, lib/pbr rgb_to_srgb,
                       geometry.py:304-318 progressive eps -- fully reference code.

Only data (inputs + expected outputs) is written; no reference source text is stored.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
import oracle  # noqa: E402


class _Anything:
    """Placeholder returned for any attribute the reference imports from an absent package."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Anything()


class _StubModule(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Anything


class _AutoStubFinder:
    """Any submodule of an absent third-party package resolves to a permissive stub."""

    ROOTS = ("matplotlib", "pytorch_lightning", "torchmetrics", "torchvision", "PIL", "cv2",
             "imageio", "pyexr", "trimesh", "mcubes", "torch_efficient_distloss", "pkg_resources",
             "nvdiffrast", "omegaconf", "lpips", "kornia", "skimage", "open3d", "tensorboard")

    def find_spec(self, name, path=None, target=None):
        import importlib.machinery
        if name.split(".")[0] in self.ROOTS and name not in sys.modules:
            return importlib.machinery.ModuleSpec(name, self)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def _stub(name, **attrs):
    m = _StubModule(name)
    m.__dict__.update(attrs)
    m.__path__ = []
    sys.modules[name] = m
    return m


class _OracleEncoding(torch.nn.Module):
    """Stands in for tcnn.Encoding(otype=HashGrid) with the oracle's hash grid."""

    def __init__(self, n_input_dims, cfg):
        super().__init__()
        self.otype = cfg["otype"]
        if self.otype == "SphericalHarmonics":
            # tcnn's SH encoding is equally absent: stand in with the oracle's (unpinned) basis
            self.n_input_dims, self.degree = n_input_dims, cfg["degree"]
            self.n_output_dims = cfg["degree"] ** 2
            return
        assert cfg["otype"] == "HashGrid"
        self.meta, n_params = oracle.grid_meta(cfg["n_levels"], cfg["n_features_per_level"],
                                               cfg["log2_hashmap_size"], cfg["base_resolution"],
                                               cfg["per_level_scale"])
        self.n_input_dims = n_input_dims
        self.n_output_dims = cfg["n_levels"] * cfg["n_features_per_level"]
        g = torch.Generator().manual_seed(7)
        self.params = torch.nn.Parameter((torch.rand(n_params, generator=g) * 2 - 1) * 1e-1)

    def forward(self, x):
        if self.otype == "SphericalHarmonics":
            from oracle import texture as otex
            return otex.sh_encode(x, self.degree)
        return oracle.hashgrid_encode(x, self.params, self.meta)


def _weight_from_alpha(alphas, ray_indices=None, n_rays=None):
    w = torch.zeros_like(alphas)
    t = torch.zeros_like(alphas)
    T, prev = 1.0, -1
    for i in range(alphas.shape[0]):
        r = int(ray_indices[i])
        if r != prev:
            T, prev = 1.0, r
        t[i] = T
        w[i] = alphas[i] * T
        T = T * (1.0 - float(alphas[i]))
    return w, t


def _accumulate(weights, values=None, ray_indices=None, n_rays=None):
    src = weights[:, None] if values is None else weights[:, None] * values
    return torch.zeros(n_rays, src.shape[-1]).index_add(0, ray_indices, src)


def install_stubs():
    _stub("tinycudann", Encoding=_OracleEncoding, Network=object, NetworkWithInputEncoding=object,
          free_temporary_memory=lambda: None)
    pl = _stub("pytorch_lightning", LightningModule=torch.nn.Module, LightningDataModule=object)
    plu = _stub("pytorch_lightning.utilities")
    rz = _stub("pytorch_lightning.utilities.rank_zero", rank_zero_info=lambda *a, **k: None,
               rank_zero_debug=lambda *a, **k: None, rank_zero_warn=lambda *a, **k: None)
    pl.utilities, plu.rank_zero = plu, rz

    class _OC:
        @staticmethod
        def register_new_resolver(*a, **k):
            pass

        @staticmethod
        def to_container(c, resolve=True):
            return dict(c)

    _stub("omegaconf", OmegaConf=_OC)
    nd = _stub("nvdiffrast")
    nd.torch = _stub("nvdiffrast.torch", texture=None)
    _stub("nerfacc", OccGridEstimator=object, accumulate_along_rays=_accumulate,
          render_weight_from_alpha=_weight_from_alpha, render_weight_from_density=None,
          ray_aabb_intersect=None)
    _stub("nerfacc.volrend", accumulate_along_rays=_accumulate,
          render_weight_from_alpha=_weight_from_alpha, render_weight_from_density=None,
          rendering=None)
    for name in ["imageio", "pyexr", "cv2", "torch_efficient_distloss", "torchmetrics",
                 "torchmetrics.functional", "torchmetrics.functional.image",
                 "torchmetrics.functional.image.lpips", "trimesh", "mcubes", "matplotlib",
                 "pytorch_lightning.callbacks", "pytorch_lightning.callbacks.progress",
                 "pytorch_lightning.loggers", "pytorch_lightning.loggers.base",
                 "pytorch_lightning.utilities.types", "torchvision", "torchvision.transforms",
                 "torchvision.transforms.functional", "PIL", "pkg_resources"]:
        if name not in sys.modules:
            _stub(name)
    sys.meta_path.append(_AutoStubFinder())
    sys.path.insert(0, REF)


class Cfg(dict):
    """Minimal OmegaConf-node look-alike (attribute access, .get, .copy)."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return Cfg(v) if isinstance(v, dict) else v

    def copy(self):
        return Cfg(dict.copy(self))


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        out[k] = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name, {k: v.shape for k, v in out.items()})


def main():
    install_stubs()
    torch.manual_seed(0)
    import models  # noqa: F401  (reference package)
    from models import network_utils as nu
    from models import geometry as geo
    from models import ray_utils, volrend
    from models.split_mixed_occ import SplitMixedOCCModel
    import utils.misc as misc
    misc.get_rank = lambda: 0

    # ---- VanillaMLP --------------------------------------------------------------------
    for tag, (din, dout, nn_, nh) in {"a": (35, 48, 64, 2), "b": (11, 13, 32, 2),
                                     "c": (35, 48, 128, 2)}.items():
        torch.manual_seed(11)
        mlp = nu.VanillaMLP(din, dout, {"n_neurons": nn_, "n_hidden_layers": nh,
                                        "sphere_init": True, "sphere_init_radius": 0.5,
                                        "weight_norm": True, "output_activation": "none"})
        # perturb g and b so that weight_norm is not the identity
        with torch.no_grad():
            for p in mlp.parameters():
                p.add_(torch.randn_like(p) * 0.01)
        x = torch.rand(257, din) * 2 - 1
        x.requires_grad_(True)
        y = mlp(x)
        gy = torch.randn_like(y)
        grads = torch.autograd.grad(y, [x] + list(mlp.parameters()), gy)
        sd = {k.replace(".", "_"): v for k, v in mlp.state_dict().items()}
        gnames = ["gx"] + ["grad_" + n.replace(".", "_") for n, _ in mlp.named_parameters()]
        save(f"vanilla_mlp_{tag}.npz", x=x, y=y, gy=gy, dims=np.array([din, dout, nn_, nh]),
             **sd, **dict(zip(gnames, grads)))

    # ---- ReLU MLP (texture-style, kaiming init, no weight norm) ---------------------------
    torch.manual_seed(12)
    mlp = nu.VanillaMLP(20, 3, {"n_neurons": 64, "n_hidden_layers": 2, "output_activation": "none"})
    x = torch.randn(130, 20)
    save("vanilla_mlp_relu.npz", x=x, y=mlp(x),
         **{k.replace(".", "_"): v for k, v in mlp.state_dict().items()})

    # ---- get_alpha ---------------------------------------------------------------------
    class _V(torch.nn.Module):
        def __init__(self, v):
            super().__init__()
            self.variance = torch.tensor(v)

        def forward(self, x):
            return torch.ones([len(x), 1]) * torch.exp(self.variance * 10.0)

    S = 513
    sdf = torch.randn(S) * 0.05
    normal = torch.nn.functional.normalize(torch.randn(S, 3), dim=-1)
    dirs = torch.nn.functional.normalize(torch.randn(S, 3), dim=-1)
    dists = torch.full((S, 1), 0.00507421875) * (1 + torch.rand(S, 1))
    out = {}
    for vi, v in enumerate([0.3, 0.55]):
        for ci, car in enumerate([1.0, 0.25]):
            fake = types.SimpleNamespace(variance=_V(v), cos_anneal_ratio=car)
            out[f"alpha_v{vi}_c{ci}"] = SplitMixedOCCModel.get_alpha(fake, sdf, normal, dirs, dists)
    save("get_alpha.npz", sdf=sdf, normal=normal, dirs=dirs, dists=dists,
         variances=np.array([0.3, 0.55]), cos_anneal=np.array([1.0, 0.25]), **out)

    # ---- VolumeSDF with FD gradient ------------------------------------------------------
    gcfg = Cfg({
        "name": "volume-sdf", "radius": 1.5, "feature_dim": 13, "grad_type": "finite_difference",
        "finite_difference_eps": "progressive", "isosurface": None,
        "xyz_encoding_config": {"otype": "ProgressiveBandHashGrid", "n_levels": 6,
                                "n_features_per_level": 2, "log2_hashmap_size": 12,
                                "base_resolution": 8, "per_level_scale": 1.5,
                                "include_xyz": True, "start_level": 3, "start_step": 0,
                                "update_steps": 100},
        "mlp_network_config": {"otype": "VanillaMLP", "activation": "ReLU",
                               "output_activation": "none", "n_neurons": 32,
                               "n_hidden_layers": 2, "sphere_init": True,
                               "sphere_init_radius": 0.5, "weight_norm": True},
    })
    nu.config_to_primitive = lambda c: dict(c)
    orig_cuda_device = torch.cuda.device

    class _NoDev:
        def __init__(self, *a):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    torch.cuda.device = _NoDev
    # ProgressiveBandHashGrid allocates its mask with device=get_rank() (== cuda:0): patch zeros
    orig_zeros = torch.zeros

    def zeros_cpu(*a, **k):
        k.pop("device", None)
        return orig_zeros(*a, **k)

    torch.zeros = zeros_cpu
    torch.manual_seed(21)
    vsdf = geo.VolumeSDF(gcfg)
    torch.zeros = orig_zeros
    torch.cuda.device = orig_cuda_device
    vsdf.contraction_type = geo.ContractionType.AABB
    vsdf.train()
    pts = (torch.rand(300, 3) * 2 - 1) * 1.5
    pts[:8] = torch.tensor([[1.5, 0, 0], [-1.5, 0.2, 0.1], [0, 1.5, 0], [0, -1.5, 0],
                            [0.3, 0.1, 1.5], [0, 0, -1.5], [1.4999, 1.4999, 1.4999], [0, 0, 0]])
    res = {}
    for step in [0, 150, 1000]:
        vsdf.update_step(0, step)
        sdf_v, grad_v, feat_v = vsdf(pts, with_grad=True, with_feature=True)
        params = list(vsdf.parameters())
        gs, gg = torch.randn_like(sdf_v), torch.randn_like(grad_v)
        loss = (sdf_v * gs).sum() + (grad_v * gg).sum() + (feat_v ** 2).sum() * 0.1
        grads = torch.autograd.grad(loss, params)
        res.update({f"s{step}_sdf": sdf_v, f"s{step}_grad": grad_v, f"s{step}_feature": feat_v,
                    f"s{step}_gs": gs, f"s{step}_gg": gg,
                    f"s{step}_eps": np.array(vsdf._finite_difference_eps),
                    f"s{step}_level": np.array(vsdf.encoding.encoding.current_level)})
        for (n, _), g in zip(vsdf.named_parameters(), grads):
            res[f"s{step}_grad__" + n.replace(".", "_")] = g
    save("volume_sdf_fd.npz", pts=pts, steps=np.array([0, 150, 1000]),
         **{"p__" + k.replace(".", "_"): v for k, v in vsdf.state_dict().items()}, **res)

    # ---- rays ------------------------------------------------------------------------------
    W = H = 16
    focal = 0.5 * W / np.tan(0.5 * 0.6911112)
    dirs_cam = ray_utils.get_ray_directions(W, H, focal, focal, W / 2, H / 2)
    c2w = torch.tensor([[0.6, -0.48, 0.64, 2.56], [0.8, 0.36, -0.48, -1.92],
                        [0.0, 0.8, 0.6, 2.4]], dtype=torch.float32)
    ro, rd = ray_utils.get_rays(dirs_cam, c2w)
    save("rays.npz", W=W, H=H, focal=focal, c2w=c2w, directions=dirs_cam, rays_o=ro, rays_d=rd)

    # ---- rendering_with_normals_sdf orchestration ---------------------------------------------
    ri = torch.tensor([0, 0, 0, 2, 2, 3, 3, 3, 3])
    ts = torch.tensor([0.1, 0.2, 0.3, 0.5, 0.6, 0.1, 0.2, 0.3, 0.4])
    te = ts + 0.1
    S = ri.numel()
    rgbs, nrm = torch.rand(S, 7), torch.randn(S, 3)
    alphas, sdf_s, sdfg = torch.rand(S), torch.randn(S), torch.randn(S, 3)
    c, n, o, d, ex = volrend.rendering_with_normals_sdf(
        ts, te, ray_indices=ri, n_rays=5,
        rgb_alpha_fn=lambda a, b, c_: (rgbs, nrm, alphas, sdf_s, sdfg), color_dim=7)
    save("rendering.npz", ray_indices=ri, t_starts=ts, t_ends=te, rgbs=rgbs, normals_in=nrm,
         alphas=alphas, colors=c, normals=n, opacities=o, depths=d, weights=ex["weights"],
         trans=ex["trans"])

    # ---- VolumeMixedMipSplitOcc.forward, stage 0 (models/texture.py:292-327) --------------------------
    # Reference code for the blend / activation logic, the five VanillaMLPs and VanillaFrequency; the SH
    # direction encoding is the oracle's stand-in for tcnn (unpinned).  The FG LUT file is not in the
    # repo (README.md:68) and is not touched at stage 0: np.fromfile is patched to return zeros.
    from models import texture as rtex
    orig_fromfile = np.fromfile
    np.fromfile = lambda *a, **k: np.zeros(256 * 256 * 2, dtype=np.float32)
    mlp_cfg = lambda n: {"otype": "VanillaMLP", "activation": "ReLU", "output_activation": "none",
                         "n_neurons": 64, "n_hidden_layers": n}
    tcfg = Cfg({
        "name": "volume-mixed-mip-split-occ", "input_feature_dim": 13, "other_dim": 3, "sample_size": 8,
        "dir_encoding_config": {"otype": "SphericalHarmonics", "degree": 5, "reflected": True},
        "metallic_mlp_network_config": mlp_cfg(2), "albedo_mlp_network_config": mlp_cfg(4),
        "spec_mlp_network_config": mlp_cfg(4), "roughness_mlp_network_config": mlp_cfg(2),
        "secondary_mlp_network_config": mlp_cfg(4),
        "xyz_encoding_config": {"otype": "VanillaFrequency", "n_frequencies": 6},
        "color_activation": "sigmoid",
    })
    torch.cuda.device = _NoDev
    torch.manual_seed(31)
    tex = rtex.VolumeMixedMipSplitOcc(tcfg)
    torch.cuda.device = orig_cuda_device
    np.fromfile = orig_fromfile
    S = 257
    feats = torch.randn(S, 13, requires_grad=True)
    dirs = torch.nn.functional.normalize(torch.randn(S, 3), dim=-1)
    nrm = torch.nn.functional.normalize(torch.randn(S, 3), dim=-1).requires_grad_(True)
    pos = (torch.rand(S, 3) * 2 - 1) * 1.5
    col = tex(feats, dirs, nrm, pos, None, 0)
    gcol = torch.randn_like(col)
    tparams = list(tex.parameters())
    tgrads = torch.autograd.grad(col, [feats, nrm] + tparams, gcol, allow_unused=True)
    tsd = {"p__" + k.replace(".", "_"): v for k, v in tex.state_dict().items() if k != "FG_LUT"}
    tg = {"g__" + n.replace(".", "_"): (g if g is not None else torch.zeros_like(p))
          for (n, p), g in zip(tex.named_parameters(), tgrads[2:])}
    save("texture_stage0.npz", features=feats, dirs=dirs, normals=nrm, positions=pos, colors=col, gcolors=gcol,
         g_features=tgrads[0], g_normals=tgrads[1], **tsd, **tg)

    # ---- stage 1: EnvironmentLightMipCube (lib/pbr/light.py:127-210) + split-sum shading
    # (models/texture.py:329-345).  Reference code for build_mips / get_mip / eval_mip / cubemap_mip and the
    # shading arithmetic; its CUDA-only callees are the oracle's restatements, evaluated in fp64:
    #   ru.diffuse_cubemap / ru.specular_cubemap -> oracle.envlight (renderutils plugin, unbuildable here)
    #   dr.texture                               -> oracle.envlight cube sampling / oracle.gridsample (nvdiffrast
    #                                               is absent upstream: definition unpinned)
    # The FG LUT file is not in the repository: np.fromfile returns oracle.texture.synthetic_fg_lut().
    from oracle import envlight as oenv, gridsample as ogs, texture as otex
    import nvdiffrast.torch as dr_stub

    def dr_texture(tex, uv, mip=None, mip_level_bias=None, filter_mode="linear", boundary_mode="wrap"):
        if boundary_mode == "cube":
            d = uv.reshape(-1, 3).double()
            if filter_mode == "linear":
                out = oenv.cube_sample_linear(tex[0].double(), d)
            else:
                out = oenv.cube_sample_mip([tex[0].double()] + [m[0].double() for m in mip], d,
                                           mip_level_bias.reshape(-1).double())
            return out.float().reshape(*uv.shape[:-1], -1)
        assert boundary_mode == "clamp" and filter_mode == "linear"
        grid = (uv.double() * 2.0 - 1.0)
        out = ogs.grid_sample_2d(tex.double().permute(0, 3, 1, 2), grid, "border", False)
        return out.permute(0, 2, 3, 1).float()

    dr_stub.texture = dr_texture
    from lib.pbr import light as rlight
    rlight.ru.diffuse_cubemap = lambda c: oenv.diffuse_cubemap(c.double()).float()
    rlight.ru.specular_cubemap = lambda c, r, cutoff=0.99: oenv.specular_cubemap(c.double(), r, cutoff).float()
    rlight.dr.texture = dr_texture
    from lib.pbr.utils import light_utils as rlu
    rlu.dr.texture = dr_texture
    orig_rand, orig_linspace = torch.rand, torch.linspace

    def _nodev(fn):
        def f(*a, **k):
            k.pop("device", None)
            return fn(*a, **k)
        return f

    torch.rand, torch.zeros, torch.linspace = _nodev(orig_rand), _nodev(orig_zeros), _nodev(orig_linspace)
    torch.manual_seed(41)
    lcfg = Cfg({"envlight_config": {"scale": 0.5, "bias": 0.25, "base_res": 64, "hdr_filepath": None}})
    light = rlight.EnvironmentLightMipCube(lcfg)
    light.build_mips()
    np.fromfile = lambda *a, **k: otex.synthetic_fg_lut().numpy().reshape(-1)
    torch.cuda.device = _NoDev
    torch.manual_seed(31)
    tex1 = rtex.VolumeMixedMipSplitOcc(tcfg)
    tex1.load_state_dict({k: v for k, v in tex.state_dict().items() if k != "FG_LUT"}, strict=False)
    assert float(tex1.FG_LUT.abs().max()) > 0
    torch.cuda.device = orig_cuda_device
    np.fromfile = orig_fromfile
    col1 = tex1(feats, dirs, nrm, pos, light, 1)
    assert col1.shape == (S, 24)
    gcol1 = torch.randn_like(col1)
    t1params = list(tex1.parameters())
    t1grads = torch.autograd.grad(col1, [feats, nrm, light.base] + t1params, gcol1, allow_unused=True)
    torch.rand, torch.zeros, torch.linspace = orig_rand, orig_zeros, orig_linspace
    t1g = {"g__" + n.replace(".", "_"): (g if g is not None else torch.zeros_like(p))
           for (n, p), g in zip(tex1.named_parameters(), t1grads[3:])}
    mips = {"spec%d" % i: m for i, m in enumerate(light.specular)}
    save("texture_stage1.npz", base=light.base, diffuse=light.diffuse, **mips, colors=col1, gcolors=gcol1,
         g_features=t1grads[0], g_normals=t1grads[1], g_base=t1grads[2], **t1g)

    # ---- the same two forwards at the widths that ship (configs/split-mixed-occ-tensoir.yaml:93-120: n_neurons 128,
    # input_feature_dim 48): reference-run GRADIENT fixtures for the layer-pair kernels (csrc/mlp_pair.hip).  The weights are
    # regenerated from their names on both sides (tests/helpers.py::seeded_param) and the cotangent has the dynamic range of
    # composite weights (helpers.composite_like_cotangent), so each fixture holds inputs, outputs and gradients only.
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from helpers import composite_like_cotangent, seeded_param
    rng_state = torch.get_rng_state()                     # the fixtures below this block keep their random streams
    mlp128 = lambda n: {"otype": "VanillaMLP", "activation": "ReLU", "output_activation": "none",
                        "n_neurons": 128, "n_hidden_layers": n}
    tcfg128 = Cfg({
        "name": "volume-mixed-mip-split-occ", "input_feature_dim": 48, "other_dim": 3, "sample_size": 8,
        "dir_encoding_config": {"otype": "SphericalHarmonics", "degree": 5, "reflected": True},
        "metallic_mlp_network_config": mlp128(2), "albedo_mlp_network_config": mlp128(4),
        "spec_mlp_network_config": mlp128(4), "roughness_mlp_network_config": mlp128(2),
        "secondary_mlp_network_config": mlp128(4),
        "xyz_encoding_config": {"otype": "VanillaFrequency", "n_frequencies": 6},
        "color_activation": "sigmoid",
    })
    np.fromfile = lambda *a, **k: otex.synthetic_fg_lut().numpy().reshape(-1)
    torch.cuda.device = _NoDev
    tex128 = rtex.VolumeMixedMipSplitOcc(tcfg128)
    torch.cuda.device = orig_cuda_device
    np.fromfile = orig_fromfile
    with torch.no_grad():
        for n, p_ in tex128.named_parameters():
            p_.copy_(seeded_param(n, tuple(p_.shape), seed=128))
    torch.manual_seed(51)
    S = 385                                               # 12 full 32-row tiles + a ragged one
    feats = torch.randn(S, 48, requires_grad=True)
    dirs = torch.nn.functional.normalize(torch.randn(S, 3), dim=-1)
    nrm = torch.nn.functional.normalize(torch.randn(S, 3), dim=-1).requires_grad_(True)
    pos = (torch.rand(S, 3) * 2 - 1) * 1.5
    col = tex128(feats, dirs, nrm, pos, None, 0)
    gcol = composite_like_cotangent(tuple(col.shape), seed=52)
    tgrads = torch.autograd.grad(col, [feats, nrm] + list(tex128.parameters()), gcol, allow_unused=True)
    tg = {"g__" + n.replace(".", "_"): (g if g is not None else torch.zeros_like(p_))
          for (n, p_), g in zip(tex128.named_parameters(), tgrads[2:])}
    save("texture_stage0_n128.npz", features=feats, dirs=dirs, normals=nrm, positions=pos, colors=col, gcolors=gcol,
         g_features=tgrads[0], g_normals=tgrads[1], **tg)
    torch.rand, torch.zeros, torch.linspace = _nodev(orig_rand), _nodev(orig_zeros), _nodev(orig_linspace)
    light128 = rlight.EnvironmentLightMipCube(lcfg)
    with torch.no_grad():
        light128.base.copy_(seeded_param("emitter.base", (6, 64, 64, 3), seed=128).abs() * 8.0 + 0.05)
    light128.build_mips()
    col1 = tex128(feats, dirs, nrm, pos, light128, 1)
    assert col1.shape == (S, 24)
    gcol1 = composite_like_cotangent(tuple(col1.shape), seed=53)
    t1grads = torch.autograd.grad(col1, [feats, nrm, light128.base] + list(tex128.parameters()), gcol1, allow_unused=True)
    torch.rand, torch.zeros, torch.linspace = orig_rand, orig_zeros, orig_linspace
    t1g = {"g__" + n.replace(".", "_"): (g if g is not None else torch.zeros_like(p_))
           for (n, p_), g in zip(tex128.named_parameters(), t1grads[3:])}
    save("texture_stage1_n128.npz", colors=col1, gcolors=gcol1, g_features=t1grads[0], g_normals=t1grads[1],
         g_base=t1grads[2], **t1g)
    torch.set_rng_state(rng_state)

    # ---- N3: state_dict layout of the reference modules (names, shapes) for checkpoint compatibility ----------
    import json
    from models import split_mixed_occ as rsmo
    var = rsmo.VarianceNetwork(Cfg({"init_val": 0.3, "modulate": False}))
    layout = {prefix: {k: list(v.shape) for k, v in mod.state_dict().items()}
              for prefix, mod in (("geometry", vsdf), ("texture", tex1), ("variance", var), ("emitter", light))}
    with open(os.path.join(HERE, "state_dict_layout.json"), "w") as f:
        json.dump(layout, f, indent=0, sort_keys=True)
    print("wrote state_dict_layout.json", {k: len(v) for k, v in layout.items()})

    # ---- frequency encoding, sRGB, progressive eps ----------------------------------------------
    vf = nu.VanillaFrequency(3, {"n_frequencies": 6})
    x = torch.randn(64, 3)
    from lib.pbr.utils.nvdiffrecmc_util import rgb_to_srgb
    lin = torch.rand(64, 3) * 1.2
    save("freq_srgb.npz", x=x, freq=vf(x), lin=lin, srgb=rgb_to_srgb(lin))


if __name__ == "__main__":
    main()
