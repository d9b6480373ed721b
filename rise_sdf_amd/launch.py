"""Zero-edit launcher: run the reference's own ``launch.py`` against this build.

    python -m rise_sdf_amd.launch [--per-layer] /path/to/RISE-SDF/launch.py --config configs/... --gpu 0 --train

does what ``python launch.py ...`` does from inside the reference tree, with two things put in place first, so that
launch.py itself (launch.py:40-42 ``import datasets, systems, models``) needs no edit:

1. ``rise_sdf_amd.dropin.install()``: ``nerfacc`` / ``tinycudann`` / ``nvdiffrast`` and the two in-tree JIT extensions
   resolve to the HIP-backed modules of this package;
2. unless ``--per-layer`` is given, the reference's model registry (models/__init__.py:1-14) is pre-populated with this
   package's classes under the reference's own names (``volume-sdf``, ``neus``, ``split-mixed-occ``, ...), which is what
   selects the fused stencil kernels.  The reference's modules are imported exactly as launch.py would import them (its
   directory first on ``sys.path``), then the six entries are replaced; launch.py's later ``import models`` is a cache hit.

Several GPUs (``--gpu 0,1``: launch.py:84-97 builds ``Trainer(strategy='ddp')``): Lightning's subprocess launcher starts
ranks 1..N-1 as ``python launch.py ...``, not through this module.  ``main`` therefore exports ``RSDF_LAUNCH_SCRIPT`` /
``RSDF_LAUNCH_FUSED`` and prepends ``rise_sdf_amd/_launch_hook`` to ``PYTHONPATH``; its ``sitecustomize`` runs the same
``prepare()`` in every interpreter that executes that script (tests/test_launcher.py runs a two-rank case).

The script is executed with ``runpy`` as ``__main__`` in this process: nothing is re-exec'd after a GPU may have been
touched, ``sys.argv`` is what launch.py would have seen, and its ``CUDA_VISIBLE_DEVICES`` handling (launch.py:36-38) still
runs before anything initialises the device (importing torch or this package does not).
"""
from __future__ import annotations

import os
import runpy
import sys

REGISTRY_NAMES = ("volume-sdf", "neus", "split-mixed-occ", "volume-mixed-mip-split-occ", "volume-radiance",
                  "envlight-mip-cube")


def _allocator_size_classes():
    """Full-image renders allocate per-chunk buffers whose sizes follow the chunk's sample count; without size classes the
    caching allocator's reserved memory creeps towards the whole HBM ("HBM footprint").  Same setting as
    bench.py; a PYTORCH_HIP_ALLOC_CONF / PYTORCH_CUDA_ALLOC_CONF from the environment wins."""
    if os.environ.get("PYTORCH_HIP_ALLOC_CONF") or os.environ.get("PYTORCH_CUDA_ALLOC_CONF"):
        return
    try:
        import torch
        setter = getattr(torch._C, "_accelerator_setAllocatorSettings", None) or torch.cuda.memory._set_allocator_settings
        setter("roundup_power2_divisions:4")
    except Exception:   # noqa: BLE001  (an allocator without this knob: nothing to do)
        pass


def prepare(script: str, fused: bool = True):
    """Everything ``main`` does before handing over to the script; returns the reference root."""
    from . import dropin
    script = os.path.abspath(script)
    if not os.path.isfile(script):
        raise SystemExit(f"rise_sdf_amd.launch: {script} does not exist")
    root = os.path.dirname(script)
    if root not in sys.path[:1]:
        sys.path.insert(0, root)            # `python launch.py` puts the script's directory first
    dropin.install()
    _allocator_size_classes()
    if fused:
        import importlib
        ref_models = importlib.import_module("models")        # the reference's registry module
        from .registry import models as mine
        reg = getattr(ref_models, "models", None)
        if not isinstance(reg, dict):
            raise SystemExit("rise_sdf_amd.launch: the script's `models` package has no `models` registry dict "
                             "(models/__init__.py:1-14)")
        for name in REGISTRY_NAMES:
            if name in mine:
                reg[name] = mine[name]
    return root


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    fused = True
    if argv and argv[0] == "--per-layer":
        fused, argv = False, argv[1:]
    if not argv or argv[0] in ("-h", "--help"):
        raise SystemExit(__doc__)
    script = os.path.abspath(argv[0])
    if "--gpu" in argv[1:-1]:               # launch.py:36-38 does the same; done here too because `models` is imported
        os.environ.setdefault("CUDA_DEVICE_ORDER", "PCI_BUS_ID")      # before the script body runs
        os.environ["CUDA_VISIBLE_DEVICES"] = argv[argv.index("--gpu", 1) + 1]
    prepare(script, fused)
    # child ranks (Lightning's subprocess launcher re-runs the script itself): see _launch_hook/sitecustomize.py
    hook = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_launch_hook")
    os.environ["RSDF_LAUNCH_SCRIPT"] = script
    os.environ["RSDF_LAUNCH_FUSED"] = "1" if fused else "0"
    os.environ["PYTHONPATH"] = os.pathsep.join([hook] + [p for p in os.environ.get("PYTHONPATH", "").split(os.pathsep)
                                                          if p and os.path.abspath(p) != hook])
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
