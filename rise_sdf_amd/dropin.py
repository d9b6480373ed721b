"""``rise_sdf_amd.dropin.install()``: make the reference's third-party / JIT-extension imports resolve to this
build's HIP-backed modules.  Call it before ``import datasets, systems, models`` (launch.py:40-42).

    import nerfacc, nerfacc.volrend          -> rise_sdf_amd.nerfacc           (0.5.3 call signatures)
    import tinycudann as tcnn                -> rise_sdf_amd.tinycudann
    import nvdiffrast.torch as dr            -> rise_sdf_amd.nvdiffrast.torch
    lib.nerfacc.cuda._backend._C             -> rise_sdf_amd.nerfacc.cuda      (vendored 0.3.5 extension surface)
    lib.renderutils.ops._get_plugin()        -> rise_sdf_amd.renderutils.plugin  (patched when ``lib`` is importable)

Nothing here copies or edits reference files: it only fills ``sys.modules`` slots the reference would otherwise
fill with CUDA builds."""
from __future__ import annotations

import sys

SLOTS = ("nerfacc", "nerfacc.volrend", "tinycudann", "nvdiffrast", "nvdiffrast.torch",
         "lib.nerfacc.cuda._backend")


def install(patch_renderutils: bool = True):
    from . import nerfacc, nvdiffrast, renderutils, tinycudann
    from .nerfacc import cuda as nerfacc_cuda
    sys.modules["nerfacc"] = nerfacc
    sys.modules["nerfacc.volrend"] = nerfacc.volrend
    sys.modules["tinycudann"] = tinycudann
    sys.modules["nvdiffrast"] = nvdiffrast
    sys.modules["nvdiffrast.torch"] = nvdiffrast.torch
    sys.modules["lib.nerfacc.cuda._backend"] = nerfacc_cuda._backend
    if patch_renderutils:
        try:
            import lib.renderutils.ops as ru_ops      # the reference's own module, when running inside its tree
        except Exception:                             # not inside the reference tree: nothing to patch
            return
        ru_ops._get_plugin = renderutils._get_plugin


def uninstall():
    for k in SLOTS:
        sys.modules.pop(k, None)
