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


def _register_all():
    from . import envlight, geometry, neus, split_mixed_occ, texture  # noqa: F401


_register_all()
