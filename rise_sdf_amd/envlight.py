"""Host-side mirror of the reference's environment light (``lib/pbr/light.py`` EnvironmentLightMipCube,
registered as ``envlight-mip-cube``) and of the renderutils / nvdiffrast entry points it calls, on the
HIP kernels of csrc/envlight.hip.

  diffuse_cubemap / specular_cubemap   lib/renderutils/ops.py:391-458
  cubemap_mip                          lib/pbr/utils/light_utils.py:94-109
  texture_cube (dr.texture, cube)      lib/pbr/light.py:194-206
  EnvironmentLightMipCube              lib/pbr/light.py:127-210 (build_mips, get_mip, eval_mip, parameters)
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch
import torch.nn as nn

from ._lib import check, lib, ptr, require_device, stream_ptr
from .registry import register


def _f(t):
    return t.detach().to(torch.float32).contiguous()


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])
    return arr, ctypes.cast(arr, ctypes.c_void_p)


# ---- prefilters ---------------------------------------------------------------------------------------
class _DiffuseCubemap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cubemap):
        c = _f(cubemap)
        require_device(c)
        out = torch.empty_like(c)
        check(lib().rsdf_diffuse_cubemap_fwd(ptr(c), c.shape[1], ptr(out), stream_ptr()), "diffuse_cubemap_fwd")
        ctx.R = c.shape[1]
        return out

    @staticmethod
    def backward(ctx, dout):
        g = _f(dout)
        gc = torch.empty_like(g)
        check(lib().rsdf_diffuse_cubemap_bwd(ptr(g), ctx.R, ptr(gc), stream_ptr()), "diffuse_cubemap_bwd")
        return gc


def diffuse_cubemap(cubemap):
    return _DiffuseCubemap.apply(cubemap)


def ndf_cutoff(roughness, cutoff, n_samples=1000000):
    """cos(theta) keeping ``cutoff`` of the GGX NDF energy (lib/renderutils/ops.py:428-439; host side, cached)."""
    a2 = roughness ** 4
    ct = np.cos(np.linspace(0, np.pi / 2.0, n_samples))
    c = np.clip(ct, 0.0, 1.0)
    d = (c * a2 - c) * c + 1.0
    D = np.cumsum(a2 / (d * d * np.pi))
    return float(ct[np.argmax(D >= D[-1] * cutoff)])


_bounds_cache = {}
_table_cache = {}


def specular_bounds(res, roughness, cutoff, device):
    key = (res, roughness, cutoff, str(device))
    if key not in _bounds_cache:
        cosc = ndf_cutoff(roughness, cutoff)
        b = torch.empty(6, res, res, 24, dtype=torch.float32, device=device)
        check(lib().rsdf_specular_bounds(res, cosc, ptr(b), stream_ptr()), "specular_bounds")
        _bounds_cache[key] = (cosc, b)
    return _bounds_cache[key]


def texel_table(res, device):
    """[6,R,R,4] = (unit texel direction, solid angle / 4), cached per resolution like the bounds."""
    key = (res, str(device))
    if key not in _table_cache:
        t = torch.empty(6, res, res, 4, dtype=torch.float32, device=device)
        check(lib().rsdf_cubemap_texel_table(res, ptr(t), stream_ptr()), "cubemap_texel_table")
        _table_cache[key] = t
    return _table_cache[key]


class _SpecularCubemapNormalized(torch.autograd.Function):
    """lib/renderutils/ops.py:455-458: specular_cubemap's `out[..., 0:3] / out[..., 3:]` inside the node.  The weight sum
    does not depend on the cube map, so the backward is the gather of dy / wsum: two elementwise kernels per level instead
    of the ten that autograd issues for the slices and the division (the 4096-ray training step is launch-bound)."""

    @staticmethod
    def forward(ctx, cubemap, roughness, cosc, bounds):
        c = _f(cubemap)
        require_device(c, bounds)
        R = c.shape[1]
        table = texel_table(R, c.device)
        # (round 6: the kernel normalises -- `out[..., 0:3] / out[..., 3:]` on the strided halves of a [6,R,R,4] tensor was a
        # broadcasting aten kernel of 100-180 us per level and direction, 1 ms of a 16 ms training step)
        out = torch.empty(6, R, R, 3, dtype=torch.float32, device=c.device)
        wsum = torch.empty(6, R, R, 1, dtype=torch.float32, device=c.device)
        check(lib().rsdf_specular_cubemap_fwd_norm(ptr(c), ptr(bounds), ptr(table), R, float(roughness), float(cosc),
                                                   ptr(out), ptr(wsum), stream_ptr()), "specular_cubemap_fwd")
        ctx.save_for_backward(bounds, table, wsum)
        ctx.args = (R, float(roughness), float(cosc))
        return out

    @staticmethod
    def backward(ctx, dy):
        bounds, table, wsum = ctx.saved_tensors
        R, roughness, cosc = ctx.args
        g = (dy / wsum).contiguous()
        gc = torch.empty(6, R, R, 3, dtype=torch.float32, device=g.device)
        check(lib().rsdf_specular_cubemap_bwd(ptr(g), 3, ptr(bounds), ptr(table), R, roughness, cosc, ptr(gc),
                                              stream_ptr()), "specular_cubemap_bwd")
        return gc, None, None, None


def specular_cubemap(cubemap, roughness, cutoff=0.99):
    assert cubemap.shape[0] == 6 and cubemap.shape[1] == cubemap.shape[2], \
        "Bad shape for cubemap tensor: %s" % str(cubemap.shape)
    cosc, bounds = specular_bounds(cubemap.shape[1], roughness, cutoff, cubemap.device)
    return _SpecularCubemapNormalized.apply(cubemap, roughness, cosc, bounds)


# ---- cube lookups ---------------------------------------------------------------------------------------
class _CubeSample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dirs, level, n_mips, *mips):
        d = _f(dirs)
        lv = None if level is None else _f(level).reshape(-1)
        ms = [_f(m) for m in mips]
        require_device(d, lv, *ms)
        R0, C = ms[0].shape[1], ms[0].shape[3]
        out = torch.empty(d.shape[0], C, dtype=torch.float32, device=d.device)
        arr, p = _ptr_array(ms)
        check(lib().rsdf_cube_sample_fwd(p, len(ms), R0, C, ptr(d), ptr(lv), d.shape[0], ptr(out), stream_ptr()),
              "cube_sample_fwd")
        ctx.save_for_backward(d, lv if lv is not None else torch.empty(0, device=d.device), *ms)
        ctx.has_level, ctx.dims = lv is not None, (R0, C)
        ctx.level_shape = None if level is None else level.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        d, lv, *ms = ctx.saved_tensors
        lv = lv if ctx.has_level else None
        R0, C = ctx.dims
        g = _f(dout)
        need_tex = [ctx.needs_input_grad[3 + i] for i in range(len(ms))]
        # one zero-fill for the gradients of all mip levels (a fill per level was 7 launches of a launch-bound step)
        sizes = [m.numel() if need else 0 for m, need in zip(ms, need_tex)]
        pool = torch.zeros(sum(sizes), dtype=torch.float32, device=d.device) if sum(sizes) else None
        grads, off = [], 0
        for m, need, k in zip(ms, need_tex, sizes):
            grads.append(pool[off:off + k].view(m.shape) if need else None)
            off += k
        gd = torch.empty_like(d) if ctx.needs_input_grad[0] else None
        gl = torch.empty(d.shape[0], dtype=torch.float32, device=d.device) \
            if (ctx.has_level and ctx.needs_input_grad[1]) else None
        a1, p1 = _ptr_array(ms)
        a2, p2 = _ptr_array(grads)
        check(lib().rsdf_cube_sample_bwd(p1, p2, len(ms), R0, C, ptr(d), ptr(lv), d.shape[0], ptr(g), ptr(gd),
                                         ptr(gl), stream_ptr()), "cube_sample_bwd")
        if gl is not None:
            gl = gl.view(ctx.level_shape)
        return (gd, gl, None, *grads)


def texture_cube(tex, dirs, mips=None, mip_level_bias=None):
    """dr.texture(tex[None], dirs[None,:,None,:], mip=..., mip_level_bias=..., boundary_mode='cube') for
    flat ``dirs`` [S,3]: 'linear' when ``mips`` is None, else 'linear-mipmap-linear' with level = bias."""
    stack = [tex] + (list(mips) if mips is not None else [])
    return _CubeSample.apply(dirs, mip_level_bias if mips is not None else None, len(stack), *stack)


_dirs_cache = {}


def _texel_dirs(R, device):
    """Face vectors of every texel, [6,R,R,3]; cached per resolution (22 elementwise launches otherwise, five times per step)."""
    key = (R, str(device))
    if key not in _dirs_cache:
        _dirs_cache[key] = _texel_dirs_uncached(R, device)
    return _dirs_cache[key]


def _texel_dirs_uncached(R, device):
    c = 2.0 * ((torch.arange(R, dtype=torch.float32, device=device) + 0.5) / R) - 1.0
    fy, fx = torch.meshgrid(c, c, indexing="ij")
    one = torch.ones_like(fx)
    faces = [torch.stack((one, -fy, -fx), -1), torch.stack((-one, -fy, fx), -1), torch.stack((fx, one, fy), -1),
             torch.stack((fx, -one, -fy), -1), torch.stack((fx, -fy, one), -1), torch.stack((-fx, -fy, -one), -1)]
    return torch.stack(faces, 0)


class cubemap_mip(torch.autograd.Function):
    """lib/pbr/utils/light_utils.py:94-109."""

    @staticmethod
    def forward(ctx, cubemap):
        c = _f(cubemap)
        require_device(c)
        R, C = c.shape[1], c.shape[3]
        out = torch.empty(6, R // 2, R // 2, C, dtype=torch.float32, device=c.device)
        check(lib().rsdf_cubemap_avgpool(ptr(c), R, C, ptr(out), stream_ptr()), "cubemap_avgpool")
        return out

    @staticmethod
    def backward(ctx, dout):
        res = dout.shape[1] * 2
        d = _texel_dirs(res, dout.device).reshape(-1, 3)
        with torch.no_grad():
            return texture_cube(_f(dout) * 0.25, d).reshape(6, res, res, dout.shape[-1])


@register("envlight-mip-cube")
class EnvironmentLightMipCube(nn.Module):
    LIGHT_MIN_RES = 16
    MIN_ROUGHNESS = 0.08
    MAX_ROUGHNESS = 0.5

    def __init__(self, config):
        super().__init__()
        self.config = config
        ec = config.envlight_config
        if ec.get("hdr_filepath", None) is not None:
            raise NotImplementedError("loading an HDR lat-long map is I/O outside the hot path")
        base = torch.rand(6, ec.base_res, ec.base_res, 3, dtype=torch.float32) * ec.scale + ec.bias
        self.register_parameter("base", nn.Parameter(base))
        self.specular, self.diffuse = None, None
        self._ready = None        # event of a build_mips_on() whose results the current stream has not waited for yet

    def __getstate__(self):
        d = self.__dict__.copy()        # (a pending event is not state: copy.deepcopy / pickle of a light that has run)
        d["_ready"] = None
        return d

    def build_mips_on(self, stream, cutoff=0.99):
        """build_mips() issued on ``stream`` (a side stream), ordered after everything already enqueued on the current one;
        eval_mip() makes the consuming stream wait for it.  The prefilter is vector-ALU work on an L2-resident cube map,
        the networks evaluated meanwhile are matrix / memory work: the two overlap on the CUs.  Autograd runs the backward
        of these nodes on ``stream`` as well (and joins the streams at the end of backward), so the prefilter's backward
        -- the last nodes of the graph -- overlaps the rest of the backward pass the same way.  Same values."""
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            self.build_mips(cutoff)
            self._ready = torch.cuda.Event()
            self._ready.record(stream)

    def _wait_ready(self):
        if self._ready is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(self._ready)
            # the mips were allocated on the side stream: tell the caching allocator that this stream reads them too, or a
            # later build_mips_on() could be handed their blocks while lookups enqueued here are still pending
            for t in list(self.specular or []) + ([self.diffuse] if self.diffuse is not None else []):
                if t.is_cuda:
                    t.record_stream(cur)
            self._ready = None

    def build_mips(self, cutoff=0.99):
        self.specular = [self.base]
        while self.specular[-1].shape[1] > self.LIGHT_MIN_RES:
            self.specular += [cubemap_mip.apply(self.specular[-1])]
        self.diffuse = diffuse_cubemap(self.specular[-1])
        n = len(self.specular)
        for idx in range(n - 1):
            roughness = (idx / (n - 2)) * (self.MAX_ROUGHNESS - self.MIN_ROUGHNESS) + self.MIN_ROUGHNESS
            self.specular[idx] = specular_cubemap(self.specular[idx], roughness, cutoff)
        self.specular[-1] = specular_cubemap(self.specular[-1], 1.0, cutoff)

    def get_mip(self, roughness):
        n = len(self.specular)
        return torch.where(
            roughness < self.MAX_ROUGHNESS,
            (torch.clamp(roughness, self.MIN_ROUGHNESS, self.MAX_ROUGHNESS) - self.MIN_ROUGHNESS)
            / (self.MAX_ROUGHNESS - self.MIN_ROUGHNESS) * (n - 2),
            (torch.clamp(roughness, self.MAX_ROUGHNESS, 1.0) - self.MAX_ROUGHNESS) / (1.0 - self.MAX_ROUGHNESS) + n - 2)

    def eval_mip(self, directions, specular=False, roughness=None):
        self._wait_ready()
        if specular:
            assert roughness is not None
            miplevel = self.get_mip(roughness)
            return texture_cube(self.specular[0], directions, mips=self.specular[1:],
                                mip_level_bias=miplevel[..., 0])
        return texture_cube(self.diffuse, directions)

    def parameters(self, recurse=True):
        return [self.base]

    def clamp_(self, min=None, max=None):
        self.base.data.clamp_(min, max)
