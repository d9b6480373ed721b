"""ctypes binding of librisesdf_hip.so (the C ABI declared in include/risesdf_hip.h).

The product path has no CPU fallback: if the shared library is missing, ``lib()`` raises.  Build it
with ``python __graft_entry__.py`` (or ``make -C rise_sdf_amd/csrc``).
"""
from __future__ import annotations

import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# RSDF_LIB selects another build of the same ABI (tools/ab_*.sh, tools/stamps_*.sh put their experiment builds there instead
# of overwriting the shipped library)
SO_PATH = os.environ.get("RSDF_LIB") or os.path.join(_HERE, "librisesdf_hip.so")
MAX_LEVELS = 32
ABI_VERSION = 3

ACT_NONE, ACT_RELU, ACT_SOFTPLUS100, ACT_SIGMOID = 0, 1, 2, 3
ACT_IDS = {"none": ACT_NONE, None: ACT_NONE, "relu": ACT_RELU, "softplus100": ACT_SOFTPLUS100,
           "sigmoid": ACT_SIGMOID}


class GridMeta(ctypes.Structure):
    """struct rsdf_grid_meta"""
    _fields_ = [
        ("n_levels", ctypes.c_uint32),
        ("n_features", ctypes.c_uint32),
        ("scale", ctypes.c_float * MAX_LEVELS),
        ("res", ctypes.c_uint32 * MAX_LEVELS),
        ("offset", ctypes.c_uint32 * MAX_LEVELS),
        ("size", ctypes.c_uint32 * MAX_LEVELS),
    ]


_P = ctypes.c_void_p
_I = ctypes.c_int
_L = ctypes.c_int64
_F = ctypes.c_float

# name -> argtypes (restype is int unless listed in _RESTYPES)
_SIGNATURES = {
    "rsdf_abi_version": [],
    "rsdf_last_error": [],
    "rsdf_ray_aabb_intersect": [_P, _P, _P, _L, _P, _P, _P],
    "rsdf_march_count": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _L, _P, _P],
    "rsdf_scan_scratch_bytes": [_L],
    "rsdf_pack_from_counts": [_P, _L, _P, _P, _P, _P],
    "rsdf_march_write": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _L, _P, _P, _P, _P, _P],
    "rsdf_march_count_staged": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _L, _P, _L, _P, _P, _P],
    "rsdf_march_write_staged": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _L, _P, _P, _L, _P, _P, _P, _P, _P, _P],
    "rsdf_query_occ": [_P, _P, _P, _I, _I, _I, _L, _P, _P, _P],
    "rsdf_counts_from_ray_indices": [_P, _L, _L, _P, _P],
    "rsdf_unpack_info": [_P, _L, _P, _P],
    "rsdf_compact_samples": [_P, _P, _P, _P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "rsdf_weight_from_alpha_fwd": [_P, _P, _L, _P, _P, _P],
    "rsdf_weight_from_alpha_bwd": [_P, _P, _P, _P, _P, _L, _P, _P],
    "rsdf_weight_from_alpha_bwd_seq": [_P, _P, _P, _P, _L, _I, _P, _P],
    "rsdf_transmittance_from_alpha_bwd": [_P, _P, _P, _P, _L, _P, _P],
    "rsdf_visibility_from_alpha": [_P, _P, _L, _F, _F, _P, _P],
    "rsdf_accumulate_fwd": [_P, _P, _P, _L, _I, _P, _P],
    "rsdf_accumulate_bwd": [_P, _P, _P, _P, _L, _I, _P, _P, _P],
    "rsdf_opacity_depth_fwd": [_P, _P, _P, _P, _L, _P, _P, _P, _P],
    "rsdf_opacity_depth_bwd": [_P, _P, _P, _P, _P, _L, _P, _P],
    "rsdf_opacity_depth_normal_fwd": [_P, _P, _P, _P, _P, _L, _P, _P, _P, _P, _P],
    "rsdf_opacity_depth_normal_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _L, _P, _P, _P],
    "rsdf_grid_meta_init": [ctypes.POINTER(GridMeta), _I, _I, _I, _I, ctypes.c_double],
    "rsdf_hashgrid_fwd": [_P, _P, ctypes.POINTER(GridMeta), _L, _I, _P, _I, _I, _I, _F, _F, _P],
    "rsdf_hashgrid_fwd_staged_scratch_bytes": [ctypes.POINTER(GridMeta), _L, _I],
    "rsdf_hashgrid_fwd_staged": [_P, _P, ctypes.POINTER(GridMeta), _L, _I, _P, _I, _I, _I, _F, _F, _P, _L, _P],
    "rsdf_hashgrid_bwd": [_P, _P, ctypes.POINTER(GridMeta), _L, _I, _I, _I, _P, _P],
    "rsdf_hashgrid_bwd_fd7_scratch_bytes": [ctypes.POINTER(GridMeta), _L, _I, _F],
    "rsdf_hashgrid_fwd_fd7": [_P, _P, ctypes.POINTER(GridMeta), _L, _I, _P, _P],
    "rsdf_hashgrid_bwd_fd7": [_P, _P, ctypes.POINTER(GridMeta), _L, _I, _F, _P, _P, _L, _P],
    "rsdf_hashgrid_scatter_binned_scratch_bytes": [ctypes.POINTER(GridMeta), _L, _I],
    "rsdf_hashgrid_scatter_binned": [_I, _P, _P, _I, _I, _P, ctypes.POINTER(GridMeta), _L, _I, _P, _P, _L, _P],
    "rsdf_hashgrid_fwd_fd7_pts": [_P, _F, _F, _P, ctypes.POINTER(GridMeta), _L, _I, _P, _P],
    "rsdf_hashgrid_bwd_fd7_pts": [_P, _F, _F, _P, ctypes.POINTER(GridMeta), _L, _I, _F, _P, _P, _L, _P],
    "rsdf_loss_rays_fwd": [_P, _P, _P, _P, _P, _P, _L, _P, _P],
    "rsdf_loss_rays_bwd": [_P, _P, _P, _P, _P, _P, _P, _L, _P, _P, _P, _P],
    "rsdf_loss_samples_fwd": [_P, _P, _P, _F, _L, _P, _P],
    "rsdf_loss_samples_bwd": [_P, _P, _P, _F, _P, _L, _P, _P, _P, _P],
    "rsdf_gen_rays": [_P, _L, _P, _P, _P, _I, _P, _P, _I, _P, _P, _I, _I, _I, _L, _P, _P, _P, _P],
    "rsdf_occ_cell_points": [_P, _P, _P, _I, _I, _I, _L, _P, _P],
    "rsdf_occ_update_scratch_bytes": [_L],
    "rsdf_occ_update": [_P, _P, _L, _F, _F, _L, _P, _P, _P, _P],
    "rsdf_hashgrid_dx": [_P, _P, ctypes.POINTER(GridMeta), _L, _I, _P, _I, _I, _P, _P],
    "rsdf_hashgrid_dx_bwd": [_P, _P, ctypes.POINTER(GridMeta), _L, _I, _P, _I, _I, _P, _P, _I, _I, _P, _P, _P],
    "rsdf_linear_fwd": [_P, _I, _P, _P, _L, _I, _I, _I, _P, _I, _P],
    "rsdf_linear_bwd_input": [_P, _P, _I, _P, _L, _I, _I, _I, _I, _I, _P, _P, _I, _P],
    "rsdf_linear_bwd_weight": [_P, _I, _P, _I, _L, _I, _I, _P, _P, _P],
    "rsdf_linear_bwd_fused_supported": [_I, _I],
    "rsdf_linear_bwd_fused": [_P, _P, _I, _P, _I, _P, _L, _I, _I, _I, _I, _I, _P, _I, _I, _P, _P, _P],
    "rsdf_linear_bwd_fused_tail": [_P, _I, _P, _P, _I, _P, _I, _P, _L, _I, _I, _I, _I, _I, _P, _I, _I, _P, _P, _P],
    "rsdf_linear_bwd_fused_workspace_bytes": [_L, _I, _I],
    "rsdf_linear_bwd_fused_ws": [_P, _P, _I, _P, _I, _P, _L, _I, _I, _I, _I, _I, _P, _I, _I, _P, _P, _P, _L, _P],
    "rsdf_linear_bwd_fused_tail_ws": [_P, _I, _P, _P, _I, _P, _I, _P, _L, _I, _I, _I, _I, _I, _P, _I, _I, _P, _P, _P, _L, _P],
    "rsdf_sdfmlp_fd7_supported": [_I, _I, _I],
    "rsdf_x2_rows": [_L],
    "rsdf_x2_bytes": [_L, _I],
    "rsdf_stencil_points_tap_major": [_P, _L, _P, _P],
    "rsdf_stencil_planes_to_rows": [_P, _P, _L, _I, _I, _P, _I, _I, _I, _F, _F, _P],
    "rsdf_stencil_rows_to_planes": [_P, _I, _I, _L, _I, _P, _P],
    "rsdf_hashgrid_fwd_fd7_x2": [_P, _P, _F, _F, _P, ctypes.POINTER(GridMeta), _L, _I, _F, _F, _I, _P, _P],
    "rsdf_sdfmlp_fd7_x2_supported": [_I, _I, _I],
    "rsdf_sdfmlp_fd7_fwd_x2": [_P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _L, _P, _P, _P, _P, _P],
    "rsdf_sdfmlp_fd7_bwd_x2": [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _L, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P,
                               _P, _P, _P, _P],
    "rsdf_pair_supported": [_I, _I, _I],
    "rsdf_pair_image_bytes": [_L],
    "rsdf_pair_pack": [_P, _I, _I, _L, _P, _P, _P],
    "rsdf_pair_pack2": [_P, _I, _I, _P, _I, _I, _L, _P, _P, _P],
    "rsdf_pair_unpack": [_P, _L, _P, _P],
    "rsdf_pair_fwd": [_P, _I, _P, _P, _P, _P, _L, _P, _P, _P, _P, _I, _I, _P, _P, _P],
    "rsdf_set_record_format": [_I],
    "rsdf_get_record_format": [],
    "rsdf_pair_fwd16": [_P, _I, _P, _P, _P, _P, _L, _P, _P, _P, _P, _I, _I, _P, _P, _P],
    "rsdf_pair_bwd16": [_P, _I, _P, _P, _P, _P, _L, _P, _I, _P, _P, _P, _I, _P, _P, _P, _I, _I, _P, _I, _I, _I, _P, _P, _P, _P, _P,
                        _P],
    "rsdf_pair_bound_from_rows": [_P, _L, _P, _P],
    "rsdf_pair_bound_from_out_layer": [_P, _L, _I, _P, _P, _P, _P],
    "rsdf_pair_bwd": [_P, _I, _P, _P, _P, _P, _L, _P, _I, _P, _P, _P, _I, _P, _P, _P, _I, _I, _P, _I, _I, _I, _P, _P, _P, _P, _P,
                      _P],
    "rsdf_sdfmlp_fd7_fwd": [_P, _P, _I, _I, _F, _F, _I, _I, _P, _P, _P, _P, _P, _P, _L, _P, _P, _P, _P],
    "rsdf_sdfmlp_fd7_bwd": [_P, _P, _I, _I, _F, _F, _I, _I, _P, _P, _P, _P, _P, _P, _L, _P, _P, _P, _P,
                            _P, _P, _P, _P, _P, _P, _P],
    "rsdf_weight_norm_fwd": [_P, _P, _I, _I, _P, _P],
    "rsdf_weight_norm_bwd": [_P, _P, _P, _I, _I, _P, _P, _P],
    "rsdf_fd_points": [_P, _P, _P, _P, _P, _L, _F, _F, _P, _P, _I, _P],
    "rsdf_fd_taps": [_P, _L, _F, _F, _P, _P],
    "rsdf_fd_gradient_fwd": [_P, _I, _F, _L, _P, _P, _P],
    "rsdf_fd_gradient_bwd": [_P, _P, _F, _L, _P, _I, _P],
    "rsdf_neus_alpha_fd_fwd": [_P, _I, _P, _P, _P, _P, _P, _F, _F, _L, _P, _P, _P, _P, _P],
    "rsdf_neus_alpha_fd_bwd": [_P, _I, _P, _P, _P, _P, _P, _F, _F, _L, _P, _P, _P, _P, _P, _I, _P, _P],
    "rsdf_freq_encode": [_P, _L, _I, _F, _F, _P, _P, _I, _I, _P],
    "rsdf_sh_encode_fwd": [_P, _L, _I, _P, _I, _I, _P],
    "rsdf_sh_encode_bwd": [_P, _P, _L, _I, _I, _I, _P, _P],
    "rsdf_reflect_fwd": [_P, _P, _L, _P, _P, _P],
    "rsdf_reflect_bwd": [_P, _P, _L, _P, _P, _P, _P],
    "rsdf_split_color0_fwd": [_P, _P, _P, _L, _P, _P],
    "rsdf_split_color0_bwd": [_P, _P, _P, _P, _L, _P, _P, _P, _P],
    "rsdf_rgb_to_srgb_fwd": [_P, _L, _P, _P],
    "rsdf_rgb_to_srgb_bwd": [_P, _P, _L, _P, _P],
    "rsdf_softplus100_slope_fwd": [_P, _L, _P, _P, _P],
    "rsdf_softplus100_slope_bwd": [_P, _P, _P, _L, _P, _P],
    "rsdf_compose_srgb_fwd": [_P, _P, _P, _L, _P, _P],
    "rsdf_compose_srgb_bwd": [_P, _P, _P, _P, _L, _P, _P, _P],
    "rsdf_split_shade1_fwd": [_P, _P, _P, _P, _P, _P, _P, _L, _P, _P],
    "rsdf_split_shade1_bwd": [_P, _P, _P, _P, _P, _P, _P, _L, _P, _P, _P, _P, _P, _P, _P, _P],
    "rsdf_diffuse_cubemap_fwd": [_P, _I, _P, _P],
    "rsdf_diffuse_cubemap_bwd": [_P, _I, _P, _P],
    "rsdf_specular_bounds": [_I, _F, _P, _P],
    "rsdf_specular_cubemap_fwd": [_P, _P, _P, _I, _F, _F, _P, _P],
    "rsdf_specular_cubemap_bwd": [_P, _I, _P, _P, _I, _F, _F, _P, _P],
    "rsdf_specular_cubemap_fwd_norm": [_P, _P, _P, _I, _F, _F, _P, _P, _P],
    "rsdf_cubemap_texel_table": [_I, _P, _P],
    "rsdf_cubemap_avgpool": [_P, _I, _I, _P, _P],
    "rsdf_cube_sample_fwd": [_P, _I, _I, _I, _P, _P, _L, _P, _P],
    "rsdf_cube_sample_bwd": [_P, _P, _I, _I, _I, _P, _P, _L, _P, _P, _P, _P],
    "rsdf_grid_sample2d_fwd": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P],
    "rsdf_grid_sample2d_bwd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "rsdf_grid_sample2d_bwd2": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P],
    "rsdf_neus_alpha_fwd": [_P, _P, _P, _P, _P, _F, _L, _P, _P],
    "rsdf_neus_alpha_bwd": [_P, _P, _P, _P, _P, _F, _L, _P, _P, _P, _P, _P],
    "rsdf_neus_occ_alpha": [_P, _P, _F, _L, _P, _P],
}
# config[4]'s bf16 MLP mode: the MLP entry points a second time with the suffix _bf16 (same signatures; split_bf16.h)
BF16_ENTRY_POINTS = ("rsdf_linear_fwd", "rsdf_linear_bwd_input", "rsdf_linear_bwd_weight",
                     "rsdf_linear_bwd_fused_supported", "rsdf_linear_bwd_fused", "rsdf_linear_bwd_fused_tail",
                     "rsdf_linear_bwd_fused_ws", "rsdf_linear_bwd_fused_tail_ws",
                     "rsdf_sdfmlp_fd7_supported", "rsdf_sdfmlp_fd7_fwd", "rsdf_sdfmlp_fd7_bwd")
for _n in BF16_ENTRY_POINTS:
    _SIGNATURES[_n + "_bf16"] = _SIGNATURES[_n]
PRECISIONS = ("fp32", "bf16")


def mlp_fn(name, precision="fp32"):
    """The MLP entry point ``name`` at ``precision``: 'fp32' (fp32-equivalent split products, the default) or 'bf16'
    (one bf16 product per k-step, fp32 accumulate; opt-in, BASELINE.json configs[4])."""
    if precision in (None, "fp32", "f32", "float32"):
        return getattr(lib(), name)
    if precision in ("bf16", "bfloat16", "fp16", "half"):     # ('fp16': the fused SDF node's own 16-bit form, fused.x2_parts;
        #                                                           everything per-layer of such a network uses the bf16 build)
        assert name in BF16_ENTRY_POINTS, name
        return getattr(lib(), name + "_bf16")
    raise ValueError(f"unknown MLP precision {precision!r} (fp32, bf16 or fp16)")


_RESTYPES = {"rsdf_last_error": ctypes.c_char_p, "rsdf_scan_scratch_bytes": ctypes.c_int64,
             "rsdf_x2_rows": ctypes.c_int64, "rsdf_x2_bytes": ctypes.c_int64, "rsdf_pair_image_bytes": ctypes.c_int64,
             "rsdf_grid_meta_init": ctypes.c_int64,
             "rsdf_hashgrid_bwd_fd7_scratch_bytes": ctypes.c_int64,
             "rsdf_hashgrid_scatter_binned_scratch_bytes": ctypes.c_int64,
             "rsdf_hashgrid_fwd_staged_scratch_bytes": ctypes.c_int64,
             "rsdf_linear_bwd_fused_workspace_bytes": ctypes.c_int64,
             "rsdf_occ_update_scratch_bytes": ctypes.c_int64}

EXPORTS = tuple(_SIGNATURES)

_lib = None


class RiseSdfHipError(RuntimeError):
    pass


def lib():
    """The loaded library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RiseSdfHipError(
                f"{SO_PATH} is missing: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C rise_sdf_amd/csrc). "
                "rise_sdf_amd has no CPU fallback.")
        l = ctypes.CDLL(SO_PATH)
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(l, name)
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, ctypes.c_int)
        if l.rsdf_abi_version() != ABI_VERSION:
            raise RiseSdfHipError("librisesdf_hip.so ABI version mismatch")
        rec = os.environ.get("RSDF_REC", "")
        if rec:      # the hash backward's queue records: 'fp32' values or the default block-float pairs ('bf20')
            if rec not in ("fp32", "bf20"):
                raise RiseSdfHipError(f"RSDF_REC={rec!r}: fp32 or bf20")
            l.rsdf_set_record_format(1 if rec == "fp32" else 0)
        _lib = l
    if _timer is not None:
        return _TimedLib(_lib, _timer)
    return _lib


class KernelTimer:
    """Optional per-entry-point timing with HIP events on the launch stream (torch's current
    stream, which is where every rsdf_* call enqueues).  Used by bench.py for the roofline object;
    never active unless installed with ``set_timer``."""

    def __init__(self):
        self.records = []  # (name, scalar args, start event, end event)

    def summary(self):
        """name -> dict(calls, ms, args=[...]) after a device synchronise."""
        out = {}
        for name, args, e0, e1 in self.records:
            d = out.setdefault(name, {"calls": 0, "ms": 0.0, "args": []})
            d["calls"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["args"].append(args)
        return out


class _TimedLib:
    def __init__(self, l, timer):
        self._l, self._t = l, timer

    def __getattr__(self, name):
        fn = getattr(self._l, name)
        # host-only queries launch nothing: not timed
        if not name.startswith("rsdf_") or "_supported" in name or name.endswith("_scratch_bytes") or \
                name in ("rsdf_last_error", "rsdf_abi_version", "rsdf_grid_meta_init"):
            return fn
        timer = self._t

        def timed(*args):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args)
            e1.record()
            timer.records.append((name, tuple(a for a in args if isinstance(a, (int, float))), e0, e1))
            return rc
        return timed


_timer = None


def set_timer(timer):
    """Install (or with None remove) a KernelTimer."""
    global _timer
    _timer = timer


_DEBUG_SYNC = os.environ.get("RSDF_DEBUG_SYNC", "")      # a file path: every entry point is logged there, then waited for


def check(rc: int, what: str):
    if _DEBUG_SYNC:
        # debug aid for GPU memory faults (they abort the process from another thread, asynchronously): name the entry point
        # BEFORE waiting for it, so that the last line of the file is the launch that faulted
        with open(_DEBUG_SYNC, "a") as f:
            f.write(what + "\n")
        torch.cuda.synchronize()
    if rc != 0:
        msg = lib().rsdf_last_error()
        raise RiseSdfHipError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


class _Ptr(ctypes.c_void_p):
    """A device pointer that keeps its tensor alive for as long as the pointer object lives, i.e. until the entry point it
    is an argument of has returned (= has enqueued its kernels): ``ptr(make_scratch())`` with a temporary used to free the
    tensor BEFORE the launch -- harmless under the caching allocator's stream-ordered reuse, a dangling pointer in
    principle, and a GPU memory fault under the guard-page allocator of the test suite."""
    _keep = None


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    p = _Ptr(t.data_ptr())
    p._keep = t
    return p


def stream_ptr():
    """The current torch HIP stream as a hipStream_t: kernels are enqueued where torch's are
    (the reference ops use at::cuda::getCurrentCUDAStream(), ray_marching.cu:235)."""
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# ---- transient workspaces -----------------------------------------------------------------------------------------------
# The fused SDF field's backward needs three large buffers that live only inside one backward call: the gradient planes
# (896 B per sample), the hash backward's record queues (~1.9 KB per sample) and d(h2) (4 H B per sample) -- 48 of the ~88
# GB a 28672-ray chunk allocates.  Their sizes follow the chunk's sample count, which differs from chunk to chunk, and
# torch's caching allocator cannot reuse a cached 12.9 GB block for a 13.0 GB request: over the 23 chunks of a view its
# reserved memory grew to 257-276 GiB of the 288 (measured: bench.py ``secondary.*.hbm_gib``), one variant of the default
# bench run ended in an out-of-memory error after a size change elsewhere, and expandable segments are not available on this
# platform.  These buffers therefore come from one growing arena per (tag, device, stream): kernels of a stream run in
# order, so the next backward call on that stream may overwrite them.  Only buffers that are produced AND consumed inside one
# autograd call may use it (the fused field's backward, and the reroute scratch of its range guard) -- anything saved for a
# later call stays with the allocator.  The arenas live OUTSIDE the caching allocator's reach (empty_cache() cannot reclaim a
# tensor that is referenced), so: an arena shrinks when a request needs less than a quarter of it (a training step after a
# full-view render), all arenas are released and the request retried when torch runs out of memory, and
# ``free_workspaces()`` (the drop-in's tcnn.free_temporary_memory()) drops them.  The key carries the host thread: two
# threads enqueueing on one stream would otherwise alias each other's buffers between their kernel sequences.
_WORKSPACES = {}
_WORKSPACE_SMALL = {}
_SHRINK_AFTER = int(os.environ.get("RSDF_WS_SHRINK_AFTER", "64"))
_WORKSPACE_EVENTS = {"grown": 0, "shrunk": 0, "released_to_device": 0}


def workspace(tag: str, nbytes: int, device) -> "torch.Tensor":
    """-> uint8 [nbytes] view of the (tag, device, current stream, host thread) arena, valid until the next request with this
    key."""
    import threading
    import torch
    dev = torch.device(device)
    key = (tag, dev.index if dev.index is not None else torch.cuda.current_device(), int(stream_ptr().value or 0),
           threading.get_ident())
    buf = _WORKSPACES.get(key)
    # an arena shrinks only after _SHRINK_AFTER consecutive requests below a quarter of it: call sites that alternate between
    # a large and a small request under one tag (primary / secondary rays of one step) would otherwise free and re-request the
    # large block every step, and the caching allocator splits the freed block for the small one -- measured as 87 -> 127 GiB
    # reserved at the headline configuration
    small = buf is not None and buf.numel() > (64 << 20) and nbytes < buf.numel() // 4
    _WORKSPACE_SMALL[key] = _WORKSPACE_SMALL.get(key, 0) + 1 if small else 0
    if buf is None or buf.numel() < nbytes or _WORKSPACE_SMALL[key] >= _SHRINK_AFTER:
        if buf is not None:
            _WORKSPACE_EVENTS["grown" if buf.numel() < nbytes else "shrunk"] += 1
        _WORKSPACE_SMALL[key] = 0
        old = _WORKSPACES.pop(key, None)
        big = old is not None and old.numel() >= (1 << 30)
        _WORKSPACE_EVENTS["released_to_device"] += int(big)
        old = buf = None                                      # (released before the new one is requested)
        if big:
            # a GiB-sized arena that is being replaced goes back to the DEVICE, not to the caching allocator's free list: no
            # later request has its size, so the cached block would only double the footprint (rare: a few times per run)
            torch.cuda.empty_cache()
        # headroom for the next, slightly larger request: a quarter for small arenas; 1/64 above 1 GiB, where a re-request is
        # cheap against the kernels that fill the arena and a quarter is gigabytes (the headline's 31 GiB of hash-backward
        # queues: 39 -> 32 GiB held; the arenas reach their final size within the first rendered view)
        want = nbytes + (nbytes // 64 if nbytes >= (1 << 30) else nbytes // 4) + (1 << 20)
        try:
            buf = torch.empty(want, dtype=torch.uint8, device=dev)
        except torch.OutOfMemoryError:
            free_workspaces()                                 # the other arenas are only scratch: give them back and retry
            torch.cuda.empty_cache()
            buf = torch.empty(want, dtype=torch.uint8, device=dev)
        _WORKSPACES[key] = buf
    return buf[:nbytes]


def workspace_f32(tag: str, shape, device) -> "torch.Tensor":
    n = 1
    for d in shape:
        n *= int(d)
    return workspace(tag, 4 * n, device).view(__import__("torch").float32).view(*shape)


def workspace_stats() -> dict:
    """{tag: GiB held} summed over devices / streams / threads, and the arena count (bench.py reports it)."""
    out, n = {}, 0
    for (tag, *_), buf in _WORKSPACES.items():
        out[tag] = round(out.get(tag, 0.0) + buf.numel() / 2.0 ** 30, 2)
        n += 1
    out["arenas"] = n
    out.update(_WORKSPACE_EVENTS)
    return out


def free_workspaces():
    """Releases the arenas (tcnn.free_temporary_memory() of the drop-in calls this) and the pair kernels' cached input image."""
    _WORKSPACES.clear()
    _WORKSPACE_SMALL.clear()
    try:
        from . import ops
        ops._PAIR_PACK_CACHE.clear()
    except Exception:   # noqa: BLE001  (during interpreter shutdown / partial import)
        pass


# ---- sticky device-side status words (include/risesdf_hip.h RSDF_STATUS_*) ------------------------------------------------
# Kernels never synchronise; what only the device can see is counted into one int32 [STATUS_WORDS] tensor per device and read
# by the host whenever it blocks on the device anyway (the marcher's sample count, a compaction count, the training step's
# two sampler counts): ``poll_status`` right behind such a read costs one more 32-byte copy on an already drained stream.
STATUS_WORDS = 8
ST_X2_FWD_NONFINITE, ST_X2_BWD_REROUTED, ST_X2_BWD_GUARDED, ST_PAIR_PACK_NONFINITE, ST_PAIR_FWD_NONFINITE = 0, 1, 2, 3, 4
_STATUS = {}
_STATUS_SEEN = {}        # device index -> the raw words at the last poll (the device-side words are MONOTONIC: never cleared)
_STATUS_TOTALS = {"x2_bwd_rerouted": 0, "x2_bwd_guarded": 0, "x2_fwd_nonfinite_total": 0, "range_reroutes": 0}

# What a forward range violation of the two-part fp16 kernels does (the reference's fp32 MLPs have no such limit and simply
# continue, models/network_utils.py:109-157):
#   "reroute" (default)  the offending kernel family -- the fused x2 SDF field and / or the radiance networks' layer pairs --
#                        is switched to the range-free kernels (three bf16 parts, fp32's exponent range) for the rest of
#                        the process, with one RuntimeWarning; the poll returns ``rerouted_now`` so that a caller holding
#                        the inputs can recompute (the samplers, rise_sdf_amd.guarded, TrainStep's skipped step);
#   "raise"              RSDF_RANGE_ERROR=raise: RiseSdfHipError naming the bounds and RSDF_X2=0 / RSDF_PAIR=0 (round 5).
_RANGE_FREE = {"x2": False, "pair": False}


def range_policy() -> str:
    return "raise" if os.environ.get("RSDF_RANGE_ERROR", "reroute") == "raise" else "reroute"


def range_free(kind: str) -> bool:
    """True once a forward range violation switched kernel family ``kind`` ('x2' or 'pair') to the range-free kernels."""
    return _RANGE_FREE[kind]


def reset_range_free():
    """Back to the default kernel families (tests; a run that reloaded a saner checkpoint)."""
    _RANGE_FREE["x2"] = _RANGE_FREE["pair"] = False


def status(device) -> "torch.Tensor":
    """The device's status words (created zeroed on first use).  Kernels only ever atomicAdd to them and the host never
    writes them again: a count that lands while another stream is being polled cannot be lost."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    t = _STATUS.get(idx)
    if t is None:
        t = torch.zeros(STATUS_WORDS, dtype=torch.int32, device=torch.device("cuda", idx))
        _STATUS[idx] = t
        _STATUS_SEEN[idx] = [0] * STATUS_WORDS
    return t


def _status_delta(idx, vals):
    """Counts since the last poll of device ``idx`` (wrap-around safe), and remember ``vals`` as seen."""
    seen = _STATUS_SEEN.setdefault(idx, [0] * STATUS_WORDS)
    d = [((int(v) - int(s)) + (1 << 31)) % (1 << 32) - (1 << 31) for v, s in zip(vals, seen)]
    _STATUS_SEEN[idx] = [int(v) for v in vals]
    return d


def poll_status(device=None, raise_on_error=True) -> dict:
    """Reads the status words of ``device`` (default: every device that has any; one small device-to-host copy: call it where
    the host blocks anyway) and returns the running totals plus what was counted since the last poll
    (``x2_fwd_nonfinite``, ``rerouted_now``).  A forward range violation since the last poll switches the offending kernel
    family to the range-free kernels (``range_policy``; RSDF_RANGE_ERROR=raise: RiseSdfHipError instead)."""
    keys = list(_STATUS) if device is None else [torch.device(device).index if torch.device(device).index is not None
                                                 else torch.cuda.current_device()]
    bad, sites, rerouted = 0, [0, 0], False
    for idx in keys:
        t = _STATUS.get(idx)
        if t is not None:
            r = consume_status(t, t.tolist(), raise_on_error=False, _defer=True)
            bad += r["x2_fwd_nonfinite"]
            sites = [a + b for a, b in zip(sites, r["_sites"])]
    return _status_result(bad, raise_on_error, sites)


def consume_status(t, vals, raise_on_error=True, _defer=False) -> dict:
    """``vals``: the words of status tensor ``t`` as the caller has just read them together with its own counts (the
    capacity-mode sampler concatenates them into its one host read per step)."""
    d = _status_delta(t.device.index, vals)
    _STATUS_TOTALS["x2_bwd_rerouted"] += d[ST_X2_BWD_REROUTED]
    _STATUS_TOTALS["x2_bwd_guarded"] += d[ST_X2_BWD_GUARDED]
    sites = d[ST_PAIR_PACK_NONFINITE:ST_PAIR_FWD_NONFINITE + 1]
    if _defer:
        return {"x2_fwd_nonfinite": d[ST_X2_FWD_NONFINITE], "_sites": sites}
    return _status_result(d[ST_X2_FWD_NONFINITE], raise_on_error, sites)


def _range_message(bad, sites):
    where = (f"{bad - sites[0] - sites[1]} in the fused SDF field, {sites[0]} in the radiance networks' input pack, "
             f"{sites[1]} in their layer pairs")
    return (f"kernels of the two-part fp16 (x2) number format produced non-finite outputs ({bad} tiles since the last check: "
            f"{where}): "
            "an operand left the format's range -- fused SDF field (csrc/mlp_x2.hip): |hash feature| or |xyz| >= 255, "
            "|effective weight| >= 1023 or a hidden activation >= 454; radiance-network layer pairs (csrc/mlp_pair.hip): "
            "|input|, |weight| or a hidden activation >= 1023 -- where the reference's fp32 MLPs stay finite.")


def _status_result(bad, raise_on_error, sites=(0, 0)):
    rerouted = False
    if bad:
        _STATUS_TOTALS["x2_fwd_nonfinite_total"] += bad
        if raise_on_error and range_policy() == "raise":
            raise RiseSdfHipError(
                _range_message(bad, sites) + "  Set RSDF_X2=0 (SDF network) / RSDF_PAIR=0 (radiance networks) to run on the "
                "range-free kernels (three bf16 parts, fp32's exponent range).")
        if raise_on_error:
            import warnings
            kinds = [k for k, n in (("x2", bad - sites[0] - sites[1]), ("pair", sites[0] + sites[1])) if n > 0 and not _RANGE_FREE[k]]
            for k in kinds:
                _RANGE_FREE[k] = True
            if kinds:
                rerouted = True
                _STATUS_TOTALS["range_reroutes"] += 1
                warnings.warn(_range_message(bad, sites) + "  Switched " + " and ".join(
                    {"x2": "the SDF field (as RSDF_X2=0)", "pair": "the radiance networks (as RSDF_PAIR=0)"}[k] for k in kinds)
                    + " to the range-free kernels (three bf16 parts, fp32's exponent range) for the rest of this process; the "
                    "call that overflowed is recomputed where its caller still holds the inputs (samplers, "
                    "rise_sdf_amd.guarded, TrainStep skips that optimizer step).  RSDF_RANGE_ERROR=raise raises instead.",
                    RuntimeWarning, stacklevel=3)
    return dict(_STATUS_TOTALS, x2_fwd_nonfinite=bad, rerouted_now=rerouted)


def status_totals() -> dict:
    """Totals accumulated by the polls so far (no device access)."""
    return dict(_STATUS_TOTALS)


def require_device(*tensors):
    """Same contract as the reference's CHECK_INPUT (helpers_cuda.h:20-25): device + contiguous."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RiseSdfHipError("rise_sdf_amd ops only accept device (HIP) tensors; there is no "
                                  "CPU path")
        if not t.is_contiguous():
            raise RiseSdfHipError("rise_sdf_amd ops need contiguous tensors")


def make_grid_meta(n_levels, n_features, log2_hashmap_size, base_resolution, per_level_scale):
    """Host-side level table via the library's own rsdf_grid_meta_init. Returns (meta, n_params)."""
    m = GridMeta()
    n_params = lib().rsdf_grid_meta_init(ctypes.byref(m), int(n_levels), int(n_features),
                                         int(log2_hashmap_size), int(base_resolution),
                                         float(per_level_scale))
    if n_params < 0:
        raise RiseSdfHipError("rsdf_grid_meta_init rejected the configuration")
    return m, int(n_params)
