"""Oracle for the environment-light rows.  TEST INFRASTRUCTURE ONLY.

  cube_to_dir / pixel_area / diffuse / specular prefilter   lib/renderutils/c_src/cubemap.cu:17-46,110-350 and
                                                            lib/renderutils/ops.py:428-458 (pinned by restatement:
                                                            the CUDA source is in the tree but needs cuda/torch
                                                            headers, unbuildable here)
  cubemap_mip                                               lib/pbr/utils/light_utils.py:94-109
  build_mips / get_mip / eval_mip                           lib/pbr/light.py:169-206
  cube texture sampling                                     nvdiffrast ``dr.texture(boundary_mode='cube')`` is
                                                            absent upstream => PARITY UNPINNED.  Definition used
                                                            here and in csrc/envlight.hip: major-axis face
                                                            selection with the face orientation of cube_to_dir,
                                                            texel centres at (i+1/2)/R, bilinear taps that leave
                                                            the face are re-projected through their 3-D direction
                                                            onto the neighbouring face (nearest texel); explicit
                                                            mip stack, level = mip_level_bias, linear between
                                                            the two nearest levels.
  split-sum shading (stage 1)                               models/texture.py:329-345
Dense O(texels^2) forms: use small resolutions.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def face_dir(s, fx, fy):
    """Unnormalised direction of face s at face coordinates (fx, fy) in [-1,1] (cubemap.cu:32-46)."""
    one = torch.ones_like(fx)
    return [torch.stack((one, -fy, -fx), -1), torch.stack((-one, -fy, fx), -1), torch.stack((fx, one, fy), -1),
            torch.stack((fx, -one, -fy), -1), torch.stack((fx, -fy, one), -1), torch.stack((-fx, -fy, -one), -1)][s]


def texel_dirs(R, dtype=torch.float32):
    """[6,R,R,3] normalised directions of texel centres; index [s, y, x]."""
    c = 2.0 * ((torch.arange(R, dtype=dtype) + 0.5) / R) - 1.0
    fy, fx = torch.meshgrid(c, c, indexing="ij")
    d = torch.stack([face_dir(s, fx, fy) for s in range(6)], 0)
    return d / d.norm(dim=-1, keepdim=True)


def pixel_area(R, dtype=torch.float32):
    """[R,R] solid-angle weights (cubemap.cu:17-30): separable atan differences."""
    if R <= 1:
        return torch.ones(R, R, dtype=dtype)
    H = R // 2
    i = (torch.arange(R) - H).abs().to(dtype)
    a = torch.atan((i + 1) / H) - torch.atan(i / H)
    return a[:, None] * a[None, :]   # [y, x]


def diffuse_cubemap(cubemap):
    """cubemap [6,R,R,3] -> [6,R,R,3]: sum_L clamp(N.L, 0, 0.999) * area / 3.141592 * c(L)  (:110-139)."""
    R = cubemap.shape[1]
    D = texel_dirs(R, cubemap.dtype).reshape(-1, 3)
    w = (D @ D.T).clamp(0.0, 0.999) * pixel_area(R, cubemap.dtype).reshape(1, -1).repeat(1, 6) / 3.141592
    return (w @ cubemap.reshape(-1, 3)).reshape(6, R, R, 3)


def ndf_cutoff(roughness, cutoff=0.99, n_samples=1000000):
    """cos(theta) below which the GGX NDF lobe holds ``cutoff`` of its energy (ops.py:428-442)."""
    a2 = roughness ** 4
    ct = np.cos(np.linspace(0, np.pi / 2.0, n_samples))
    c = np.clip(ct, 0.0, 1.0)
    d = (c * a2 - c) * c + 1.0
    D = np.cumsum(a2 / (d * d * np.pi))
    return float(ct[np.argmax(D >= D[-1] * cutoff)])


_WEIGHTS = {}      # (R, roughness, cutoff, dtype, cos_shift) -> weights: the dense matrix of a 64^2 map is 4.8 GB of fp64 and
                   # a minute of trigonometry; the fixtures of several tests share the same three levels (read-only)


def specular_weights(R, roughness, cutoff=0.99, dtype=torch.float64, chunk=1024, cos_shift=0.0):
    """Dense [6RR, 6RR] prefilter weights (row = output texel V, column = input texel L): texels with
    L.V >= cos_cutoff get (L.V) * D_ggx(V.H) * area / 4  (:246-298).  Cached per argument tuple (up to 3 entries)."""
    key = (int(R), float(roughness), float(cutoff), dtype, float(cos_shift))
    if key not in _WEIGHTS:
        while len(_WEIGHTS) >= 3:
            _WEIGHTS.pop(next(iter(_WEIGHTS)))
        _WEIGHTS[key] = _specular_weights(R, roughness, cutoff, dtype, chunk, cos_shift)
    return _WEIGHTS[key]


def _specular_weights(R, roughness, cutoff, dtype, chunk, cos_shift):
    D = texel_dirs(R, dtype).reshape(-1, 3)
    area = pixel_area(R, dtype).reshape(1, -1).repeat(1, 6)[0]
    cosc = ndf_cutoff(roughness, cutoff) + cos_shift
    a2 = (roughness * roughness) ** 2
    rows = []
    for i in range(0, D.shape[0], chunk):
        V = D[i:i + chunk]
        dot = V @ D.T
        # the NDF only where the cone test passes (a sharp lobe keeps a fraction of a per cent of the pairs): the same
        # elementwise arithmetic on the same operands as the dense form, zeros elsewhere
        vi, li = torch.nonzero(dot >= cosc, as_tuple=True)
        Vs, Ls = V[vi], D[li]
        Hh = Vs + Ls
        Hh = Hh / Hh.norm(dim=-1, keepdim=True).clamp_min(1e-20)
        vh = (Hh * Vs).sum(-1).clamp(0.0, 1.0)
        dd = (vh * a2 - vh) * vh + 1.0
        w = torch.zeros_like(dot)
        w[vi, li] = dot[vi, li].clamp_min(0.0) * (a2 / (dd * dd * math.pi)) * area[li] / 4.0
        rows.append(w)
    return torch.cat(rows, 0)


def _row_candidates(D, V, cosc, R, B):
    """(vi, li) pairs that may pass ``V[vi] . D[li] >= cosc``: B x B texel blocks whose centre direction is within the
    cone's angle + the block's own angular radius of V (a conservative cull; the exact test follows on the pairs)."""
    nb = R // B
    Db = D.reshape(6, nb, B, nb, B, 3).permute(0, 1, 3, 2, 4, 5).reshape(6 * nb * nb, B * B, 3)
    cen = Db.mean(1)
    cen = cen / cen.norm(dim=-1, keepdim=True)
    rad = torch.acos((Db * cen[:, None, :]).sum(-1).clamp(-1.0, 1.0)).max(1).values          # [blocks]
    ang = torch.acos((V @ cen.T).clamp(-1.0, 1.0))                                            # [rows, blocks]
    vi, bi = torch.nonzero(ang <= math.acos(max(min(cosc, 1.0), -1.0)) + rad[None, :] + 1e-6, as_tuple=True)
    # texel ids of block bi
    s_, by, bx = bi // (nb * nb), (bi // nb) % nb, bi % nb
    oy, ox = torch.meshgrid(torch.arange(B), torch.arange(B), indexing="ij")
    li = (s_[:, None] * R * R + (by[:, None] * B + oy.reshape(1, -1)) * R + bx[:, None] * B + ox.reshape(1, -1))
    return vi[:, None].expand_as(li).reshape(-1), li.reshape(-1)


def _face_vec_xy(s_idx, x, y, R, dtype):
    """cube_to_dir(x, y, s, R) (cubemap.cu:32-46) for index tensors (x, y may be R: one past the face, as the bounds kernel's
    tile corners are)."""
    fx = 2.0 * ((x.to(dtype) + 0.5) / R) - 1.0
    fy = 2.0 * ((y.to(dtype) + 0.5) / R) - 1.0
    one = torch.ones_like(fx)
    comps = [(one, -fy, -fx), (-one, -fy, fx), (fx, one, fy), (fx, -one, -fy), (fx, -fy, one), (-fx, -fy, -one)]
    d = torch.zeros(fx.shape + (3,), dtype=dtype)
    for k, c in enumerate(comps):
        m = s_idx == k
        if bool(m.any()):
            d[m] = torch.stack(c, -1)[m]
    return d / d.norm(dim=-1, keepdim=True)


def _tile_pass(V, R, cosc, dtype, TILE=16):
    """SpecularBoundsKernel's tile culling (cubemap.cu:201-219): per (row V, face, 16 x 16 tile) whether the "blunt interval
    arithmetic" test max(minx Vx, maxx Vx) + ... >= cutoff passes, with the component intervals taken from the tile's FOUR
    CORNER directions (corners at tile_end = one texel past the tile).  The test is not conservative -- a direction
    component can peak inside a tile -- and the reference's window is what it lets through: in-cone texels of a culled
    tile are dropped unless the bounding box of the surviving ones covers them.  -> (pass [rows, 6 nt nt] bool,
    margin [rows] = smallest |maxdp - cutoff| of the row, nt)."""
    nt = (R + TILE - 1) // TILE
    s_idx, ty, tx = torch.meshgrid(torch.arange(6), torch.arange(nt), torch.arange(nt), indexing="ij")
    s_idx, ty, tx = s_idx.reshape(-1), ty.reshape(-1), tx.reshape(-1)
    tsx, tsy = tx * TILE, ty * TILE
    tex, tey = torch.clamp((tx + 1) * TILE, max=R), torch.clamp((ty + 1) * TILE, max=R)
    corners = torch.stack([_face_vec_xy(s_idx, tsx, tsy, R, dtype), _face_vec_xy(s_idx, tex, tsy, R, dtype),
                           _face_vec_xy(s_idx, tsx, tey, R, dtype), _face_vec_xy(s_idx, tex, tey, R, dtype)], 0)
    lo, hi = corners.min(0).values, corners.max(0).values                    # [tiles, 3]
    maxdp = torch.maximum(lo[None] * V[:, None, :], hi[None] * V[:, None, :]).sum(-1)      # [rows, tiles]
    return maxdp >= cosc, (maxdp - cosc).abs().min(1).values, nt


def specular_rows(cubemap, roughness, rows, cutoff=0.99, cos_shifts=(0.0,), chunk=256, model_bounds=True,
                  return_margin=False, border=None):
    """The GGX prefilter restricted to the output texels ``rows`` (flat indices into [6RR]): the same elementwise
    arithmetic as ``_specular_weights`` on the same operands, but only the window members of the selected rows are ever
    formed, so the sizes the path runs every step -- R = 512 / 256 / 128, lib/pbr/light.py:177-180 -- fit the CPU.
    ``model_bounds``: the window is the reference's -- in-cone texels inside the per-face bounding box that
    SpecularBoundsKernel builds from the tiles its culling test lets through (cubemap.cu:181-244; see _tile_pass) -- which at
    narrow lobes (R = 512: the cone spans ~1.7 tiles) is a proper subset of the cone for some texels; False = the whole cone
    (what the dense forms above evaluate; identical at the small resolutions they are used at).
    One result [len(rows), 3] per entry of ``cos_shifts`` (the tests bracket the fp32 window compare), differentiable in
    ``cubemap`` (the backward is the transpose of the same sparse rows: cubemap.cu:300-350).  ``return_margin``: also the
    rows' smallest |tile test - cutoff| (a row within fp32 rounding of flipping a whole tile).  ``border``: also ``slack``
    [len(rows), 3], a bound on what the texels with |L.V - cutoff| <= border can move the row's output by."""
    R = cubemap.shape[1]
    dtype = cubemap.dtype
    B = 16 if R >= 64 else (4 if R % 4 == 0 else 1)
    with torch.no_grad():
        D = texel_dirs(R, dtype).reshape(-1, 3)
        area = pixel_area(R, dtype).reshape(1, -1).repeat(1, 6)[0]
        cosc0 = ndf_cutoff(roughness, cutoff)
        a2 = (roughness * roughness) ** 2
        vis, lis, ws, dots, keeps, margins = [], [], [], [], [], []
        for i in range(0, len(rows), chunk):
            V = D[rows[i:i + chunk]]
            low = min(min(cos_shifts), -(border or 0.0))
            vi, li = _row_candidates(D, V, cosc0 + low, R, B)
            Vs, Ls = V[vi], D[li]
            dot = (Vs * Ls).sum(-1)
            keep = dot >= cosc0 + low
            vi, li, Vs, Ls, dot = vi[keep], li[keep], Vs[keep], Ls[keep], dot[keep]
            Hh = Vs + Ls
            Hh = Hh / Hh.norm(dim=-1, keepdim=True).clamp_min(1e-20)
            vh = (Hh * Vs).sum(-1).clamp(0.0, 1.0)
            dd = (vh * a2 - vh) * vh + 1.0
            ws.append(dot.clamp_min(0.0) * (a2 / (dd * dd * math.pi)) * area[li] / 4.0)
            in_box = []
            if model_bounds:
                passed, margin, nt = _tile_pass(V, R, cosc0, dtype)
                margins.append(margin)
                face, ly, lx = li // (R * R), (li // R) % R, li % R
                tile = (face * nt + ly // 16) * nt + lx // 16
                for sh in tuple(cos_shifts) + ((-border,) if border is not None else ()):
                    # bounding box per (row, face) of the in-cone texels of the tiles that pass (:222-236) ...
                    m = passed[vi, tile] & (dot >= cosc0 + sh)
                    key = vi * 6 + face
                    n_key = V.shape[0] * 6
                    big = torch.iinfo(torch.int64).max
                    mnx = torch.full((n_key,), big).scatter_reduce(0, key[m], lx[m], "amin")
                    mxx = torch.full((n_key,), -1).scatter_reduce(0, key[m], lx[m], "amax")
                    mny = torch.full((n_key,), big).scatter_reduce(0, key[m], ly[m], "amin")
                    mxy = torch.full((n_key,), -1).scatter_reduce(0, key[m], ly[m], "amax")
                    # ... and every in-cone texel inside it (:263-272)
                    in_box.append((lx >= mnx[key]) & (lx <= mxx[key]) & (ly >= mny[key]) & (ly <= mxy[key]))
            else:
                in_box = [torch.ones_like(dot, dtype=torch.bool) for _ in range(len(cos_shifts) + (border is not None))]
            keeps.append(torch.stack(in_box, 0))
            vis.append(vi + i)
            lis.append(li)
            dots.append(dot)
        vi, li, w, dot, keepm = torch.cat(vis), torch.cat(lis), torch.cat(ws), torch.cat(dots), torch.cat(keeps, 1)
    outs = []
    for k, sh in enumerate(cos_shifts):
        m = (dot >= cosc0 + sh) & keepm[k]
        wsum = torch.zeros(len(rows), dtype=dtype).index_add_(0, vi[m], w[m])
        col = torch.zeros(len(rows), 3, dtype=dtype).index_add(0, vi[m], w[m][:, None] * cubemap.reshape(-1, 3)[li[m]])
        outs.append(col / wsum[:, None])
    if border is not None:
        # what the texels within ``border`` of the cutoff can move the row's output by, each taken alone and summed (two
        # borderline texels of opposite effect cancel in a lo / hi bracket, not in this bound): w_b |c_b - out| / wsum over them,
        # with the smallest window's wsum
        with torch.no_grad():
            mb = ((dot - cosc0).abs() <= border) & keepm[-1]      # (the box of the WIDEST window: a borderline texel can extend it)
            m_in = (dot >= cosc0 + border) & keepm.all(0)
            wmin = torch.zeros(len(rows), dtype=dtype).index_add_(0, vi[m_in], w[m_in])
            eff = w[mb][:, None] * (cubemap.detach().reshape(-1, 3)[li[mb]] - outs[0].detach()[vi[mb]]).abs()
            slack = torch.zeros(len(rows), 3, dtype=dtype).index_add_(0, vi[mb], eff) / wmin.clamp_min(1e-300)[:, None]
        extra = (slack,)
    else:
        extra = ()
    if return_margin:
        return (outs, (torch.cat(margins) if margins else torch.full((len(rows),), float("inf"), dtype=dtype))) + extra
    return outs if not extra else (outs,) + extra


def specular_cubemap(cubemap, roughness, cutoff=0.99, cos_shift=0.0):
    """GGX prefilter (:246-298 + ops.py:458): weighted sum normalised by the weight sum; differentiable
    in ``cubemap``.  ``cos_shift`` moves the window threshold (tests bracket the fp32 threshold compare of a
    texel whose L.V is within rounding of the cutoff)."""
    R = cubemap.shape[1]
    with torch.no_grad():
        w = specular_weights(R, roughness, cutoff, cubemap.dtype, cos_shift=cos_shift)
    col = w @ cubemap.reshape(-1, 3)
    return (col / w.sum(-1, keepdim=True)).reshape(6, R, R, 3)


# ---- cube texture sampling (definition in the module docstring) ----------------------------------------------
def dir_to_face_uv(d):
    """d [S,3] -> (face int64 [S], fx, fy in [-1,1])."""
    ax = d.abs()
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    is_x = (ax[:, 0] >= ax[:, 1]) & (ax[:, 0] >= ax[:, 2])
    is_y = (~is_x) & (ax[:, 1] >= ax[:, 2])
    face = torch.where(is_x, torch.where(x > 0, 0, 1), torch.where(is_y, torch.where(y > 0, 2, 3),
                                                                  torch.where(z > 0, 4, 5)))
    ma = torch.where(is_x, ax[:, 0], torch.where(is_y, ax[:, 1], ax[:, 2])).clamp_min(1e-30)
    fx = torch.stack([-z, z, x, x, x, -x], 0).gather(0, face[None])[0] / ma
    fy = torch.stack([-y, -y, z, -z, -y, -y], 0).gather(0, face[None])[0] / ma
    return face, fx, fy


def _fetch(tex, face, xi, yi):
    """Texel fetch with cube wrap: taps outside [0,R) are re-projected through their direction."""
    R = tex.shape[1]
    inside = (xi >= 0) & (xi < R) & (yi >= 0) & (yi < R)
    fx = 2.0 * ((xi.to(tex.dtype) + 0.5) / R) - 1.0
    fy = 2.0 * ((yi.to(tex.dtype) + 0.5) / R) - 1.0
    d = torch.stack([face_dir(s, fx, fy) for s in range(6)], 0).gather(
        0, face[None, :, None].expand(1, -1, 3))[0]
    f2, gx, gy = dir_to_face_uv(d)
    x2 = torch.floor((gx + 1.0) * 0.5 * R).clamp(0, R - 1).long()
    y2 = torch.floor((gy + 1.0) * 0.5 * R).clamp(0, R - 1).long()
    f = torch.where(inside, face, f2)
    xx = torch.where(inside, xi, x2)
    yy = torch.where(inside, yi, y2)
    return tex[f, yy, xx]


def cube_sample_linear(tex, dirs):
    """tex [6,R,R,C], dirs [S,3] (need not be normalised) -> [S,C]; differentiable in tex and dirs."""
    R = tex.shape[1]
    face, fx, fy = dir_to_face_uv(dirs)
    px = (fx + 1.0) * 0.5 * R - 0.5
    py = (fy + 1.0) * 0.5 * R - 0.5
    x0, y0 = torch.floor(px.detach()).long(), torch.floor(py.detach()).long()
    tx, ty = (px - x0)[:, None], (py - y0)[:, None]
    return (_fetch(tex, face, x0, y0) * (1 - tx) * (1 - ty) + _fetch(tex, face, x0 + 1, y0) * tx * (1 - ty)
            + _fetch(tex, face, x0, y0 + 1) * (1 - tx) * ty + _fetch(tex, face, x0 + 1, y0 + 1) * tx * ty)


def cube_sample_mip(mips, dirs, level):
    """mips: list of [6,R_l,R_l,C]; level [S] float (mip_level_bias) -> [S,C], linear-mipmap-linear."""
    n = len(mips)
    lv = level.clamp(0.0, float(n - 1))
    l0 = torch.floor(lv.detach()).clamp(max=n - 1).long()
    l1 = (l0 + 1).clamp(max=n - 1)
    t = (lv - l0)[:, None]
    out = 0
    for l in range(n):
        s = cube_sample_linear(mips[l], dirs)
        out = out + s * ((l0 == l)[:, None] * (1 - t)) + s * (((l1 == l) & (l1 != l0))[:, None] * t)
    return out


# ---- light.py / light_utils.py ----------------------------------------------------------------------------------
class _CubemapMip(torch.autograd.Function):
    """lib/pbr/utils/light_utils.py:94-109: forward = 2x2 average pool; backward = cube-linear lookup of
    0.25*dout at the finer level's texel directions (NOT the exact adjoint -- restated as the reference)."""

    @staticmethod
    def forward(ctx, cubemap):
        s, R = cubemap.shape[0], cubemap.shape[1]
        return cubemap.reshape(s, R // 2, 2, R // 2, 2, -1).mean(dim=(2, 4))

    @staticmethod
    def backward(ctx, dout):
        res = dout.shape[1] * 2
        d = texel_dirs(res, dout.dtype).reshape(-1, 3)
        return cube_sample_linear(dout * 0.25, d).reshape(6, res, res, -1)


def cubemap_mip(cubemap):
    return _CubemapMip.apply(cubemap)


LIGHT_MIN_RES, MIN_ROUGHNESS, MAX_ROUGHNESS = 16, 0.08, 0.5


def build_mips(base, cutoff=0.99):
    """lib/pbr/light.py:169-180 -> (specular list, diffuse)."""
    spec = [base]
    while spec[-1].shape[1] > LIGHT_MIN_RES:
        spec.append(cubemap_mip(spec[-1]))
    diffuse = diffuse_cubemap(spec[-1])
    for i in range(len(spec) - 1):
        r = (i / (len(spec) - 2)) * (MAX_ROUGHNESS - MIN_ROUGHNESS) + MIN_ROUGHNESS
        spec[i] = specular_cubemap(spec[i], r, cutoff)
    spec[-1] = specular_cubemap(spec[-1], 1.0, cutoff)
    return spec, diffuse


def get_mip(roughness, n_mips):
    """lib/pbr/light.py:182-185."""
    return torch.where(roughness < MAX_ROUGHNESS,
                       (roughness.clamp(MIN_ROUGHNESS, MAX_ROUGHNESS) - MIN_ROUGHNESS)
                       / (MAX_ROUGHNESS - MIN_ROUGHNESS) * (n_mips - 2),
                       (roughness.clamp(MAX_ROUGHNESS, 1.0) - MAX_ROUGHNESS) / (1.0 - MAX_ROUGHNESS) + n_mips - 2)
