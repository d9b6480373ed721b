"""Device buffers that END flush against an unmapped page (test infrastructure, GPU only).

A kernel that reads a few bytes past the end of an input usually gets away with it: torch's caching allocator hands out
pieces of large mapped segments, so the over-read lands in somebody else's (mapped) bytes.  Once in a while the buffer is the
last piece of a segment and the process dies with a GPU memory fault that no test reproduces (round 5: the per-wave SDF
forward after tests/test_gpu_ops.py had shaped the allocator's free lists).  `guarded(...)` makes that placement the
rule: every tensor it returns is mapped through HIP's virtual memory management API (hipMemAddressReserve / hipMemCreate /
hipMemMap) into a reservation whose next page is NOT mapped, with its last byte < 16 bytes from that page.

A GPU memory fault aborts the process (from a runtime thread; nothing to catch), so the cases that use these buffers run in a
child process (tests/test_gpu_guard_pages.py) and a fault shows up as a non-zero exit code of the named case.
"""
import ctypes

import numpy as np
import torch

_hip = None
_KEEP = []          # reservations live until the process exits (a test process; nothing is unmapped)


class _Location(ctypes.Structure):
    _fields_ = [("type", ctypes.c_int), ("id", ctypes.c_int)]


class _AllocFlags(ctypes.Structure):
    _fields_ = [("compressionType", ctypes.c_ubyte), ("gpuDirectRDMACapable", ctypes.c_ubyte), ("usage", ctypes.c_ushort)]


class _Prop(ctypes.Structure):      # hipMemAllocationProp (hip_runtime_api.h)
    _fields_ = [("type", ctypes.c_int), ("requestedHandleType", ctypes.c_int), ("location", _Location),
                ("win32HandleMetaData", ctypes.c_void_p), ("allocFlags", _AllocFlags)]


class _AccessDesc(ctypes.Structure):
    _fields_ = [("location", _Location), ("flags", ctypes.c_int)]


def _rt():
    global _hip
    if _hip is None:
        torch.cuda.init()
        _hip = ctypes.CDLL("libamdhip64.so")      # the copy torch has already loaded
    return _hip


def _ok(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with hipError {rc}")


def _prop(device):
    p = _Prop()
    p.type = 1                      # hipMemAllocationTypePinned
    p.requestedHandleType = 0       # hipMemHandleTypeNone
    p.location.type = 1             # hipMemLocationTypeDevice
    p.location.id = device
    return p


class _Raw:
    """What torch.as_tensor wraps without a copy."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False), "version": 2}


def guarded(shape, dtype=torch.float32, device=0, fill=None, slack=0):
    """An uninitialised (or `fill`ed) tensor of `shape` whose last byte lies `slack` + (< 16) bytes before an unmapped page.
    The base address keeps 16-byte alignment (the kernels' documented requirement for their vector loads)."""
    hip = _rt()
    shape = (shape,) if isinstance(shape, int) else tuple(shape)
    itemsize = torch.empty((), dtype=dtype).element_size()
    nbytes = int(np.prod(shape, dtype=np.int64)) * itemsize
    prop = _prop(device)
    gran = ctypes.c_size_t()
    _ok(hip.hipMemGetAllocationGranularity(ctypes.byref(gran), ctypes.byref(prop), 0), "hipMemGetAllocationGranularity")
    g = gran.value
    mapped = max(g, (nbytes + slack + 15 + g - 1) // g * g)
    base = ctypes.c_void_p()
    # one unmapped granule behind the mapping: reserved (nobody else can map there), never mapped
    _ok(hip.hipMemAddressReserve(ctypes.byref(base), ctypes.c_size_t(mapped + g), ctypes.c_size_t(g), None,
                                 ctypes.c_ulonglong(0)), "hipMemAddressReserve")
    handle = ctypes.c_void_p()
    _ok(hip.hipMemCreate(ctypes.byref(handle), ctypes.c_size_t(mapped), ctypes.byref(prop), ctypes.c_ulonglong(0)),
        "hipMemCreate")
    _ok(hip.hipMemMap(base, ctypes.c_size_t(mapped), ctypes.c_size_t(0), handle, ctypes.c_ulonglong(0)), "hipMemMap")
    acc = _AccessDesc()
    acc.location.type = 1
    acc.location.id = device
    acc.flags = 3                   # hipMemAccessFlagsProtReadWrite
    _ok(hip.hipMemSetAccess(base, ctypes.c_size_t(mapped), ctypes.byref(acc), ctypes.c_size_t(1)), "hipMemSetAccess")
    end = base.value + mapped - slack
    ptr = (end - nbytes) // 16 * 16
    if nbytes == 0:
        return torch.empty(shape, dtype=dtype, device=f"cuda:{device}")
    typestr = {torch.float32: "<f4", torch.float16: "<f2", torch.int32: "<i4", torch.int64: "<i8", torch.uint8: "|u1",
               torch.float64: "<f8", torch.int16: "<i2", torch.bool: "|b1"}[dtype]
    raw = _Raw(ptr, shape, typestr)
    t = torch.as_tensor(raw, device=f"cuda:{device}")
    assert t.data_ptr() == ptr, "torch copied the guarded buffer"
    _KEEP.append((base.value, mapped, handle, raw))
    if fill is not None:
        t.copy_(fill.to(t.device).reshape(shape)) if torch.is_tensor(fill) else t.fill_(fill)
    return t


def guarded_like(t, **kw):
    """A guarded copy of `t` (same shape, dtype and values)."""
    return guarded(t.shape, t.dtype, t.device.index or 0, fill=t.contiguous(), **kw)
