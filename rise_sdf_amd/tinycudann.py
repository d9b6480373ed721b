"""Drop-in for the tiny-cuda-nn surface RISE-SDF touches (``import tinycudann as tcnn``):
``tcnn.Encoding(n_input_dims, encoding_config)`` (models/network_utils.py:50,99) and
``tcnn.free_temporary_memory()`` (models/utils.py:120).

tiny-cuda-nn is neither vendored nor version-pinned by the reference (README.md:56), so the grid
definition is this build's own (rise_sdf_amd/csrc/hashgrid.hip; DESIGN.md "Hash grid").  Differences
from upstream tcnn that a user should know: parameters and outputs are fp32 (tcnn computes in fp16
with fp32 master weights); d(output)/d(input) is not provided (the finite-difference-normal
configurations of RISE-SDF never request it).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib, ops


class Encoding(nn.Module):
    """``tcnn.Encoding(n_input_dims, encoding_config, seed=1337, dtype=None)``: constructing it returns the
    implementation selected by ``encoding_config['otype']`` (a subclass, so ``isinstance(e, tcnn.Encoding)``
    holds and ``params`` sits directly on the module as in tcnn's state_dict).  Common surface:
    ``n_input_dims``, ``n_output_dims``, ``params``, ``dtype``, ``forward(x [S, n_input_dims]) -> [S, n_output_dims]``
    (fp32)."""

    def __new__(cls, n_input_dims=3, encoding_config=None, *args, **kwargs):
        if cls is Encoding:
            otype = str(dict(encoding_config or {}).get("otype", "HashGrid")).lower()
            if otype not in _OTYPES:
                raise NotImplementedError(f"tcnn.Encoding otype={dict(encoding_config or {}).get('otype')!r} is not "
                                          "on the RISE-SDF hot path (implemented: HashGrid, SphericalHarmonics)")
            cls = _OTYPES[otype]
        return super().__new__(cls)


class HashGridEncoding(Encoding):
    """``otype: HashGrid`` (aliases Grid/hashgrid).  One flat fp32 ``params`` Parameter laid out
    [level][entry][feature], initialised U(-1e-4, 1e-4) like tcnn's grid encodings."""

    def __init__(self, n_input_dims: int, encoding_config: dict, seed: int = 1337, dtype=None, device=None):
        super().__init__()
        cfg = dict(encoding_config)
        if n_input_dims != 3:
            raise NotImplementedError("HashGrid is implemented for 3-D inputs")
        self.otype = "HashGrid"
        self.n_input_dims = 3
        self.n_levels = int(cfg.get("n_levels", 16))
        self.n_features_per_level = int(cfg.get("n_features_per_level", 2))
        self.log2_hashmap_size = int(cfg.get("log2_hashmap_size", 19))
        self.base_resolution = int(cfg.get("base_resolution", 16))
        self.per_level_scale = float(cfg.get("per_level_scale", 2.0))
        self.n_output_dims = self.n_levels * self.n_features_per_level
        self.meta, n_params = _lib.make_grid_meta(self.n_levels, self.n_features_per_level,
                                                  self.log2_hashmap_size, self.base_resolution,
                                                  self.per_level_scale)
        gen = torch.Generator().manual_seed(seed)
        init = (torch.rand(n_params, generator=gen, dtype=torch.float32) * 2 - 1) * 1e-4
        if device is None and torch.cuda.is_available():
            device = torch.device("cuda", torch.cuda.current_device())
        self.params = nn.Parameter(init.to(device) if device is not None else init)
        self.dtype = torch.float32

    def forward(self, x: torch.Tensor, n_active_levels=None) -> torch.Tensor:
        return ops.hashgrid_encode(x.reshape(-1, 3), self.params, self.meta,
                                   n_active_levels=n_active_levels)

    def extra_repr(self):
        return (f"HashGrid L={self.n_levels} F={self.n_features_per_level} "
                f"T=2^{self.log2_hashmap_size} base={self.base_resolution} "
                f"scale={self.per_level_scale} params={self.params.numel()}")


class SphericalHarmonicsEncoding(Encoding):
    """``otype: SphericalHarmonics`` (models/network_utils.py:98-99 with yaml ``degree: 5``): directions given
    in [0,1]^3 (tcnn maps them to [-1,1] internally; callers pre-map with ``(d+1)/2``, models/texture.py:312,348)
    -> ``degree^2`` real SH coefficients.  Parameter-free (``params`` is an empty tensor attribute).
    Differentiable in the direction."""

    def __init__(self, n_input_dims: int, encoding_config: dict, seed: int = 1337, dtype=None, device=None):
        super().__init__()
        cfg = dict(encoding_config)
        if n_input_dims != 3:
            raise NotImplementedError("SphericalHarmonics takes 3-D directions")
        self.otype = "SphericalHarmonics"
        self.degree = int(cfg.get("degree", 4))
        if not 1 <= self.degree <= 5:
            raise NotImplementedError("SphericalHarmonics degree must be in [1,5] (csrc/texture.hip)")
        self.n_input_dims = 3
        self.n_output_dims = self.degree * self.degree
        self.params = torch.zeros(0, dtype=torch.float32)   # attribute only: nothing to learn or to checkpoint
        self.dtype = torch.float32

    def forward(self, d01: torch.Tensor) -> torch.Tensor:
        from . import texture_ops
        return texture_ops.sh_encode(d01.reshape(-1, 3), self.degree)

    def extra_repr(self):
        return f"SphericalHarmonics degree={self.degree}"


_OTYPES = {"hashgrid": HashGridEncoding, "grid": HashGridEncoding,
           "sphericalharmonics": SphericalHarmonicsEncoding}


def free_temporary_memory():
    """tcnn keeps a private scratch arena; here the backward-transient workspaces of the fused field and of the binned
    table scatter do (rise_sdf_amd._lib.workspace): released here, everything else goes through torch's caching allocator."""
    from . import _lib
    _lib.free_workspaces()
    return None
