"""Host-side mirror of the reference's ``models/geometry.py`` for the SDF field on the hot path.

  contract_to_unisphere   models/geometry.py:17-29 (AABB branch)
  VolumeSDF               models/geometry.py:193-327 (``volume-sdf``)

Analytic normals (grad_type 'analytic', :224-228) and the curvature term (:246-282) are built as an explicit
forward composition of once-differentiable HIP ops (hash-grid input gradient + transposed MLP chain) instead of
``autograd.grad(create_graph=True)``; one ordinary backward then covers what the reference reaches through
tcnn's double backward.

Out of scope here: MarchingCubeHelper / isosurface (mesh export),
VolumeDensity (background model).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .nerfacc import ContractionType
from .network_utils import get_encoding, get_mlp, update_module_step
from .registry import register


def scale_anything(dat, inp_scale, tgt_scale):
    """models/utils.py:109-114."""
    if inp_scale is None:
        inp_scale = [dat.min(), dat.max()]
    dat = (dat - inp_scale[0]) / (inp_scale[1] - inp_scale[0])
    return dat * (tgt_scale[1] - tgt_scale[0]) + tgt_scale[0]


def contract_to_unisphere(x, radius, contraction_type):
    if contraction_type == ContractionType.AABB:
        return scale_anything(x, (-radius, radius), (0, 1))
    raise NotImplementedError("only ContractionType.AABB is used by the shipped configs "
                              "(models/split_mixed_occ.py:66)")


class BaseModel(nn.Module):
    """models/base.py:6-32."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.rank = torch.cuda.current_device() if torch.cuda.is_available() else 0
        self.setup()
        if self.config.get("weights", None):
            self.load_state_dict(torch.load(self.config.weights))

    def setup(self):
        raise NotImplementedError

    def update_step(self, epoch, global_step, *args):
        pass

    def regularizations(self, out):
        return {}

    @torch.no_grad()
    def export(self, export_config):
        return {}


@register("volume-sdf")
class VolumeSDF(BaseModel):
    def setup(self):
        self.n_output_dims = self.config.feature_dim
        self.radius = self.config.radius
        self.contraction_type = ContractionType.AABB  # the reference assigns this from the model
        self.encoding = get_encoding(3, self.config.xyz_encoding_config)
        self.network = get_mlp(self.encoding.n_output_dims, self.n_output_dims,
                               self.config.mlp_network_config)
        self.grad_type = self.config.grad_type
        if self.grad_type not in ("finite_difference", "analytic"):
            raise ValueError(f"Unknown grad_type={self.grad_type}")
        for key in ("sdf_activation", "feature_activation"):
            if key in self.config and str(self.config[key]).lower() not in ("none",):
                raise NotImplementedError(f"{key} other than none")
        self._finite_difference_eps = None
        self.finite_difference_eps = self.config.get("finite_difference_eps", 1e-3)
        if self.encoding.include_xyz and hasattr(self.network, "input_grad_cols"):
            # positions are not differentiated under FD normals: skip d/d(xyz) in the first layer
            self.network.input_grad_cols = (3, self.encoding.n_output_dims - 3)

    # -- the fused entry the renderer uses: 7 taps per sample straight from the ray batch ---------
    def field7_from_rays(self, rays_o, rays_d, ray_indices, t_starts, t_ends):
        """-> out7 [7S, feature_dim]: rows 7i..7i+6 are the MLP outputs at sample i's centre and its
        six clamped finite-difference taps (+x,-x,+y,-y,+z,-z)."""
        x7 = ops.fd_points(rays_o, rays_d, ray_indices, t_starts, t_ends, self.radius,
                           self._finite_difference_eps)
        return self.network(self.encoding(x7.view(-1, 3), fd7_eps_unit=self._eps_unit()))

    def fused_field_available(self):
        """True when the fused stencil kernels cover this configuration (hash grid with xyz
        pass-through + 2x{32,64} Softplus(100) MLP)."""
        from . import fused
        from .network_utils import VanillaMLP
        grid, _ = self.encoding._hash()
        net = self.network
        return (grid is not None and self.encoding.include_xyz and isinstance(net, VanillaMLP)
                and fused.supported(self.encoding.n_output_dims, net.n_neurons, self.n_output_dims,
                                    net.n_hidden_layers, net.hidden_act, net.output_act,
                                    grid.n_features_per_level))

    def sdf7_from_rays(self, rays_o, rays_d, ray_indices, t_starts, t_ends, want_feature=False):
        """Fused fast path: -> (sdf7t [7, S] tap-major SDF stencil, feature [S, feature_dim] or
        None).  Gradients flow through the SDF values only."""
        from . import fused
        grid, n_active = self.encoding._hash()
        wts = self.network.effective_weights()
        precision = getattr(self.network, "precision", "fp32")
        # the x2 kernels (H = 64 / 128) take everything from the positions: x7t is neither formed nor saved for the backward
        need_taps = not fused.use_x2(3 + 2 * grid.n_levels, wts[0][0].shape[0], wts[2][0].shape[0], precision)
        x7t, pts = ops.fd_points(rays_o, rays_d, ray_indices, t_starts, t_ends, self.radius,
                                 self._finite_difference_eps, want_positions=True, tap_major=True, want_taps=need_taps)
        # the sample midpoints o + d (t0 + t1) / 2, formed by the kernel exactly as the reference's torch expression forms
        # them (models/split_mixed_occ.py:256-257): the caller's texture / curvature query takes them from here instead of
        # five more elementwise kernels (positions carry no graph under finite-difference normals)
        self._last_points = (ray_indices, pts)
        return fused.sdf_field_fd7(
            x7t, grid.params, wts, grid.meta,
            grid.n_levels if n_active is None else n_active, self.encoding.xyz_scale,
            self.encoding.xyz_offset, self._eps_unit(), want_feature,
            points=pts, radius=self.radius, eps=self._finite_difference_eps,
            precision=precision)

    def last_points(self, ray_indices):
        """The world-space midpoints of the samples of the last ``sdf7_from_rays(…, ray_indices, …)`` call, or None when the
        last call was for another sample set (identity of the index tensor, not its values)."""
        lp = getattr(self, "_last_points", None)
        return lp[1] if lp is not None and lp[0] is ray_indices else None

    def _eps_unit(self):
        return self._finite_difference_eps / (2.0 * self.radius)

    # -- analytic gradient as a forward composition (geometry.py:224-228) ------------------------------
    def field_with_analytic_grad(self, pts):
        """pts [S,3] world -> (out [S,D], grad [S,3] = d out[:,0] / d pts).  ``grad`` is built from
        once-differentiable ops, so it can be trained through (eikonal, curvature) and, when ``pts`` carries
        a graph, differentiated w.r.t. the positions as well."""
        from .network_utils import VanillaMLP
        net, enc = self.network, self.encoding
        grid, n_active = enc._hash()
        if grid is None or not isinstance(net, VanillaMLP) or net.output_act != "none":
            raise NotImplementedError("analytic gradient: hash grid + VanillaMLP (no output activation) only")
        x = contract_to_unisphere(pts, self.radius, self.contraction_type)
        col = 3 if enc.include_xyz else 0
        h = ops.hashgrid_encode(x, grid.params, grid.meta, n_active_levels=n_active, include_xyz=enc.include_xyz,
                                xyz_scale=enc.xyz_scale, xyz_offset=enc.xyz_offset)
        wb = net.effective_weights()
        slopes = []
        for w, b in wb[:-1]:
            z = ops.linear(h, w, b, act="none", precision=net.precision)
            if net.hidden_act == "softplus100" and z.is_cuda and z.dtype == torch.float32:
                from .texture_ops import softplus100_slope
                h, sl = softplus100_slope(z)          # (one kernel each way for activation + slope and their backward)
            elif net.hidden_act == "softplus100":
                h, sl = F.softplus(z, beta=100), torch.sigmoid(100.0 * z)
            else:
                h, sl = F.relu(z), (z > 0).to(z.dtype)
            slopes.append(sl)
        w_last, b_last = wb[-1]
        out = ops.linear(h, w_last, b_last, act="none", precision=net.precision)
        # reverse sweep for output channel 0: u <- (u * act'(z_i)) @ W_i
        # (the first step's u is the constant row W_last[0]: (u * sl) @ W = sl @ (diag(W_last[0]) W) -- folded into the
        # [H,H] weight instead of an [S,H] elementwise product forward and two backward)
        u = None
        for (w, _), sl in zip(reversed(wb[:-1]), reversed(slopes)):
            if u is None:
                u = ops.linear(sl, (w * w_last[0][:, None]).t().contiguous(), None, act="none", precision=net.precision)
            else:
                u = ops.linear((u * sl).contiguous(), w.t().contiguous(), None, act="none", precision=net.precision)
        if u is None:
            u = w_last[0:1].expand(out.shape[0], -1)
        g_unit = ops.hashgrid_dx(x, grid.params, u, grid.meta, n_active, col)
        if col:
            g_unit = g_unit + u[:, :3] * enc.xyz_scale
        return out, g_unit / (2.0 * self.radius)

    def curvature(self, pts, grad, rand_directions=None):
        """geometry.py:246-282 (PermutoSDF curvature): angle / pi between the normal at x and the analytic normal
        at x + 1e-4 * tangent, tangent = normal x random direction."""
        eps = 1e-4
        if rand_directions is None:
            rand_directions = torch.rand_like(pts)
        rd = F.normalize(rand_directions, dim=-1, eps=1e-6)
        normal = F.normalize(grad, dim=-1, eps=1e-6)
        pts_d = pts + eps * torch.cross(normal, rd, dim=-1)
        _, grad_d = self.field_with_analytic_grad(pts_d)
        dot = torch.sum(normal * F.normalize(grad_d, dim=-1, eps=1e-6), dim=-1)
        return torch.acos(torch.clamp(dot, -1.0 + 1e-6, 1.0 - 1e-6)) / math.pi

    def forward(self, points, with_grad=True, with_feature=True, with_laplace=False, rand_directions=None,
                input_grad=False):
        """models/geometry.py:206-292.  points [..., 3] in world space.  ``input_grad``: the caller differentiates the
        output w.r.t. ``points`` (the secondary-ray feature query), so the first layer must return d/d(xyz) for the
        pass-through columns as well; by default those columns are skipped (positions carry no graph under
        finite-difference normals)."""
        if with_laplace:
            assert self.grad_type == "finite_difference", \
                "Laplace computation is only supported with grad_type='finite_difference'"
        shape = points.shape[:-1]
        pts = points.reshape(-1, 3)
        laplace = None
        with torch.set_grad_enabled(self.training or (with_grad and self.grad_type == "analytic")):
            if with_grad and self.grad_type == "analytic":
                feature, grad = self.field_with_analytic_grad(pts)
                sdf = feature[..., 0]
            elif with_grad:
                eps = self._finite_difference_eps
                x7 = ops.fd_taps(pts, self.radius, eps)
                out7 = self.network(self.encoding(x7.view(-1, 3), fd7_eps_unit=self._eps_unit()))
                sdf, grad = ops.fd_gradient(out7, eps)
                feature = out7.view(-1, 7, self.n_output_dims)[:, 0]
                if with_laplace:
                    laplace = self.curvature(pts, grad, rand_directions)
            else:
                x = contract_to_unisphere(pts, self.radius, self.contraction_type)
                cols = getattr(self.network, "input_grad_cols", None)
                if input_grad and cols is not None and pts.requires_grad:
                    self.network.input_grad_cols = None
                try:
                    feature = self.network(self.encoding(x))
                finally:
                    if cols is not None:
                        self.network.input_grad_cols = cols
                sdf, grad = feature[..., 0], None
        rv = [sdf.view(*shape)]
        if with_grad:
            rv.append(grad.view(*shape, 3))
        if with_feature:
            rv.append(feature.reshape(*shape, self.n_output_dims))
        if with_laplace:
            rv.append(laplace.view(*shape))
        rv = [v if self.training else v.detach() for v in rv]
        return rv[0] if len(rv) == 1 else rv

    def forward_level(self, points):
        x = contract_to_unisphere(points.reshape(-1, 3), self.radius, self.contraction_type)
        return self.network(self.encoding(x))[..., 0].view(*points.shape[:-1])

    def update_step(self, epoch, global_step):
        update_module_step(self.encoding, epoch, global_step)
        update_module_step(self.network, epoch, global_step)
        if isinstance(self.finite_difference_eps, float):
            self._finite_difference_eps = self.finite_difference_eps
        elif self.finite_difference_eps == "progressive":
            hg = self.config.xyz_encoding_config
            assert hg.otype == "ProgressiveBandHashGrid", \
                "finite_difference_eps='progressive' only works with ProgressiveBandHashGrid"
            level = min(hg.start_level + max(global_step - hg.start_step, 0) // hg.update_steps,
                        hg.n_levels)
            grid_res = hg.base_resolution * hg.per_level_scale ** (level - 1)
            self._finite_difference_eps = 2 * self.config.radius / grid_res
        else:
            raise ValueError(f"Unknown finite_difference_eps={self.finite_difference_eps}")

    def regularizations(self, out):
        return {"normal_orientation": out["normals_orientation_loss_map"].mean()} \
            if "normals_orientation_loss_map" in out else {}
