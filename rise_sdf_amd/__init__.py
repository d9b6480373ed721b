"""rise_sdf_amd -- MI355X-native implementation of RISE-SDF's ray-marched SDF rendering hot path.

Layout (only what the path needs):
  csrc/            hand-written HIP kernels for gfx950 + the C ABI (include/risesdf_hip.h)
  _lib.py, ops.py  ctypes binding and torch.autograd wrappers (PyTorch = memory/stream plumbing)
  nerfacc/, tinycudann.py, nvdiffrast/, renderutils.py, dropin.py   drop-ins for the third-party / JIT surfaces the reference calls
  network_utils.py, geometry.py, neus.py, volrend.py   host-side mirror of the reference's models/
  dist.py          ray-parallel multi-GPU helpers (RCCL gradient all-reduce)

``models.register / models.make`` of the reference (models/__init__.py:1-14) are ``register`` /
``make`` here, with the same registry names (``volume-sdf``, ``neus``, ...).
"""
from .registry import make, models, register  # noqa: F401
from .config import Config, config_to_primitive, load_yaml  # noqa: F401
from ._lib import RiseSdfHipError  # noqa: F401


def check_status(device=None) -> dict:
    """Reads the kernels' sticky status words (one small device-to-host copy; blocks on the device).  Raises
    RiseSdfHipError if the fused SDF field left the range of its two-part fp16 number format since the last check;
    returns the counters (``x2_bwd_rerouted``: backward launches the range guard ran on the range-free kernels).  The
    samplers call this themselves behind the host reads they make anyway; call it after a loop that makes none."""
    from . import _lib
    return _lib.poll_status(device)


def _register_all():
    from . import envlight, geometry, neus, split_mixed_occ, texture  # noqa: F401


_register_all()
