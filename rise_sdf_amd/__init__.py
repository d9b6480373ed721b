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
    """Reads the kernels' sticky status words (one small device-to-host copy; blocks on the device).  If a kernel of the
    two-part fp16 number format left its range since the last check, that kernel family is switched to the range-free
    kernels for the rest of the process with a RuntimeWarning (``rerouted_now`` in the result; RSDF_RANGE_ERROR=raise:
    RiseSdfHipError instead); returns the counters (``x2_bwd_rerouted``: backward launches the range guard ran on the range-free kernels).  The
    samplers call this themselves behind the host reads they make anyway; call it after a loop that makes none."""
    from . import _lib
    return _lib.poll_status(device)


def guarded(fn, device=None):
    """``fn()`` with the forward range guard of the two-part fp16 kernels closed around it: the call, ONE host read of the
    status words, and -- if an operand left the format's range in it (|weight| >= 1023, |activation| >= 454 in the SDF field / 1023 in the radiance networks, |hash feature| >= 255;
    the reference's fp32 MLPs have no such bound, models/network_utils.py:109-157) -- the call again on the range-free kernels,
    which the poll has switched on for the rest of the process.  For callers that make no host read of their own between a
    forward and the use of its outputs (the samplers and TrainStep guard themselves)."""
    from . import _lib
    _lib.poll_status(device)                         # what earlier calls left behind is theirs
    out = fn()
    for _ in range(2):                               # at most once per kernel family
        if not _lib.poll_status(device)["rerouted_now"]:
            break
        out = fn()
    return out


def _register_all():
    from . import envlight, geometry, neus, split_mixed_occ, texture  # noqa: F401


_register_all()
