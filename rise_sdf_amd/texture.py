"""Host-side mirror of the reference's ``models/texture.py`` classes on the hot path.

  VolumeMixedMipSplitOcc  models/texture.py:234-434 (``volume-mixed-mip-split-occ``): the five material
                          MLPs, SH direction encoding, frequency position encoding and the stage-0
                          radiance-field output.  Stage 1 (split-sum shading with the FG LUT and the
                          prefiltered cube map, :329-345) is the next SURVEY 8a row (S1-S4, E1).
  VolumeRadiance          models/texture.py:15-41 (``volume-radiance``, the NeuS texture).
  VanillaFrequency        models/network_utils.py:14-40.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import texture_ops as T
from .config import config_to_primitive
from .network_utils import get_mlp, update_module_step
from .registry import register


class VanillaFrequency(nn.Module):
    def __init__(self, in_channels, config):
        super().__init__()
        assert in_channels == 3
        self.N_freqs = config["n_frequencies"]
        self.n_input_dims = in_channels
        self.x_scale, self.x_offset = config.get("x_scale", 1.0), config.get("x_offset", 0.0)
        self.n_output_dims = in_channels * 2 * self.N_freqs
        self.n_masking_step = config.get("n_masking_step", 0)
        self.update_step(None, None)

    def forward(self, x):
        mask = None if bool((self.mask == 1).all()) else self.mask
        return T.freq_encode(x, self.N_freqs, self.x_scale, self.x_offset, mask)

    def update_step(self, epoch, global_step):
        if self.n_masking_step <= 0 or global_step is None:
            self.mask = torch.ones(self.N_freqs, dtype=torch.float32)
        else:
            self.mask = (1.0 - torch.cos(math.pi * (global_step / self.n_masking_step * self.N_freqs
                                                    - torch.arange(0, self.N_freqs)).clamp(0, 1))) / 2.0


def SphericalHarmonics(in_channels, config):
    """tcnn.Encoding(n, {otype: SphericalHarmonics, degree: d}) (models/network_utils.py:98-99)."""
    from . import tinycudann as tcnn
    return tcnn.Encoding(in_channels, dict(config, otype="SphericalHarmonics"))


def get_dir_or_pos_encoding(n_input_dims, config):
    """The subset of get_encoding (models/network_utils.py:91-106) the texture networks use."""
    if config.otype == "VanillaFrequency":
        return VanillaFrequency(n_input_dims, config_to_primitive(config))
    if config.otype == "SphericalHarmonics":
        return SphericalHarmonics(n_input_dims, config_to_primitive(config))
    raise NotImplementedError(f"texture encoding otype {config.otype!r}")


@register("volume-mixed-mip-split-occ")
class VolumeMixedMipSplitOcc(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.n_dir_dims = self.config.get("n_dir_dims", 3)
        self.n_pos_dims = self.config.get("n_pos_dims", 3)
        self.n_output_dims = 3
        self.dir_encoding = get_dir_or_pos_encoding(self.n_dir_dims, self.config.dir_encoding_config)
        self.xyz_encoding = get_dir_or_pos_encoding(self.n_pos_dims, self.config.xyz_encoding_config)
        f, d, x = self.config.input_feature_dim, self.dir_encoding.n_output_dims, self.xyz_encoding.n_output_dims
        # construction order = the reference's (models/texture.py:247-275) so that seeded inits agree
        self.secondary_network = get_mlp(f + self.config.other_dim + d, 3, self.config.secondary_mlp_network_config)
        self.albedo_network = get_mlp(f + x, 6, self.config.albedo_mlp_network_config)
        self.roughness_network = get_mlp(f + x, 1, self.config.roughness_mlp_network_config)
        self.env_network = get_mlp(f + d, 3, self.config.spec_mlp_network_config)
        self.metallic_network = get_mlp(f + x, 2, self.config.metallic_mlp_network_config)
        if str(self.config.get("color_activation", "sigmoid")).lower() != "sigmoid":
            raise NotImplementedError("color_activation other than sigmoid")
        # precomputed microfacet integration (models/texture.py:285-287).  The reference loads
        # load/bsdf/bsdf_256_256.bin, which is not part of its repository (README.md:68); when the file is
        # absent the table is integrated analytically (split-sum GGX, see fg_lut.py).
        from .fg_lut import load_or_build_fg_lut
        self.register_buffer("FG_LUT", load_or_build_fg_lut(self.config.get("fg_lut_path", "load/bsdf/bsdf_256_256.bin")))

    def forward(self, features, dirs, normals, positions, emitter=None, stage=0, *args):
        if dirs.shape[0] == 0:
            return torch.zeros((0, 7 if stage == 0 else 24), device=features.device)
        feats = features.reshape(-1, features.shape[-1])
        wo01, nov = T.reflect(dirs, normals)
        xyz = self.xyz_encoding(positions.reshape(-1, self.n_pos_dims))
        wo_enc = self.dir_encoding(wo01)
        # torch.cat([feats, xyz]) / torch.cat([feats, wo_enc]) of models/texture.py:299,313 are the networks' two-source inputs
        # (VanillaMLP.forward(x, x2=...): the pair kernels pack them straight into their input image; one image serves the three
        # networks that read [feats, xyz])
        from . import ops as _ops
        if stage == 0:
            albedo6 = self.albedo_network(feats, x2=xyz)
            metallic2 = self.metallic_network(feats, x2=xyz)
            spec3 = self.env_network(feats, x2=wo_enc)
            # the shared input image of the material networks is scoped to this forward (the cache pinned the input
            # rows and a 512 B / row image until the next pack, and its hit test cannot see raw-pointer writes into a reused
            # buffer); the autograd nodes keep their own reference to the image for the backward
            _ops._PAIR_PACK_CACHE.clear()
            return T.split_color0(albedo6, metallic2, spec3)
        # stage 1: split-sum shading (models/texture.py:329-345); color_activation (sigmoid) fused into the
        # last layer of each material network
        from .gridsample import fg_lut_lookup
        albedo6 = self.albedo_network(feats, out_act="sigmoid", x2=xyz)
        roughness = self.roughness_network(feats, out_act="sigmoid", x2=xyz)
        metallic2 = self.metallic_network(feats, out_act="sigmoid", x2=xyz)
        spec3 = self.env_network(feats, out_act="sigmoid", x2=wo_enc)
        _ops._PAIR_PACK_CACHE.clear()
        diffuse_light = emitter.eval_mip(normals)
        wo = wo01 * 2.0 - 1.0
        specular_light = emitter.eval_mip(wo, specular=True, roughness=roughness)
        fg_uv = torch.cat([torch.clamp(nov, min=0.0, max=1.0), torch.clamp(roughness, min=0.0, max=1.0)], -1)
        fg = fg_lut_lookup(self.FG_LUT, fg_uv)
        return T.split_shade1(albedo6, roughness, metallic2, spec3, diffuse_light, specular_light, fg)

    def secondary_shading(self, features, rays_d, *args):
        dirs_embd = self.dir_encoding(((rays_d + 1.0) / 2.0).reshape(-1, self.n_dir_dims))
        inp = torch.cat([features.reshape(-1, features.shape[-1]), dirs_embd]
                        + [a.reshape(-1, a.shape[-1]) for a in args], dim=-1)
        out = torch.sigmoid(self.secondary_network(inp))
        from . import ops as _ops
        _ops._PAIR_PACK_CACHE.clear()
        return out

    def secondary_shading_pbr(self, features, dirs, normals, positions, emitter):
        """models/texture.py:386-427: split-sum shading of the third-bounce point seen along ``dirs`` (relighting).
        The specular lobe is looked up along ``dirs`` itself (:417), not along the reflection."""
        if dirs.shape[0] == 0:
            return torch.zeros((0, 3), device=dirs.device)
        from .gridsample import fg_lut_lookup
        feats = features.reshape(-1, features.shape[-1])
        _wo01, nov = T.reflect(dirs, normals)
        inp = torch.cat([feats, self.xyz_encoding(positions.reshape(-1, self.n_pos_dims))], dim=-1)
        albedo6 = self.albedo_network(inp, out_act="sigmoid")
        roughness = self.roughness_network(inp, out_act="sigmoid")
        metallic2 = self.metallic_network(inp, out_act="sigmoid")
        from . import ops as _ops
        _ops._PAIR_PACK_CACHE.clear()
        diffuse_light = emitter.eval_mip(normals)
        specular_light = emitter.eval_mip(dirs, specular=True, roughness=roughness)
        fg_uv = torch.cat([torch.clamp(nov, min=0.0, max=1.0), torch.clamp(roughness, min=0.0, max=1.0)], -1)
        fg = fg_lut_lookup(self.FG_LUT, fg_uv)
        shaded = T.split_shade1(albedo6, roughness, metallic2, torch.zeros_like(diffuse_light), diffuse_light,
                                specular_light, fg)
        return shaded[:, 7:10] + shaded[:, 10:13]

    def update_step(self, epoch, global_step):
        update_module_step(self.dir_encoding, epoch, global_step)
        update_module_step(self.xyz_encoding, epoch, global_step)

    def regularizations(self, out):
        return {}


@register("volume-radiance")
class VolumeRadiance(nn.Module):
    """models/texture.py:15-41: rgb = act(MLP([features, SH((d+1)/2), normals]))."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.n_dir_dims = self.config.get("n_dir_dims", 3)
        self.n_output_dims = 3
        self.encoding = get_dir_or_pos_encoding(self.n_dir_dims, self.config.dir_encoding_config)
        self.n_input_dims = self.config.input_feature_dim + self.encoding.n_output_dims
        self.network = get_mlp(self.n_input_dims, self.n_output_dims, self.config.mlp_network_config)
        if str(self.config.get("color_activation", "sigmoid")).lower() != "sigmoid":
            raise NotImplementedError("color_activation other than sigmoid")

    def forward(self, features, dirs, *args):
        dirs01 = (dirs + 1.0) / 2.0
        dirs_embd = self.encoding(dirs01.reshape(-1, self.n_dir_dims))
        inp = torch.cat([features.reshape(-1, features.shape[-1]), dirs_embd]
                        + [a.reshape(-1, a.shape[-1]) for a in args], dim=-1)
        color = self.network(inp).view(*features.shape[:-1], self.n_output_dims).float()
        return torch.sigmoid(color)

    def update_step(self, epoch, global_step):
        update_module_step(self.encoding, epoch, global_step)

    def regularizations(self, out):
        return {}
