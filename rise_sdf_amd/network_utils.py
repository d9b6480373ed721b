"""Host-side mirror of the reference's ``models/network_utils.py`` (same class names, config keys
and state_dict names), computing through the HIP ops.

  ProgressiveBandHashGrid   models/network_utils.py:43-68
  CompositeEncoding         models/network_utils.py:71-88
  VanillaMLP                models/network_utils.py:109-157
  get_encoding / get_mlp    models/network_utils.py:91-106, 194-204
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import ops
from . import tinycudann as tcnn
from .config import config_to_primitive


def update_module_step(m, epoch, global_step, *args):
    """systems/utils.py:349-351."""
    if hasattr(m, "update_step"):
        m.update_step(epoch, global_step, *args)


class ProgressiveBandHashGrid(nn.Module):
    """Coarse-to-fine hash grid: levels >= current_level contribute zeros.

    The reference multiplies the full encoding by a 0/1 mask (:58-61); here the mask is the
    ``n_active_levels`` argument of the gather kernel, so masked levels are never fetched."""

    def __init__(self, in_channels, config):
        super().__init__()
        self.n_input_dims = in_channels
        encoding_config = dict(config)
        encoding_config["otype"] = "HashGrid"
        self.encoding = tcnn.Encoding(in_channels, encoding_config)
        self.n_output_dims = self.encoding.n_output_dims
        self.n_level = config["n_levels"]
        self.n_features_per_level = config["n_features_per_level"]
        self.start_level, self.start_step, self.update_steps = \
            config["start_level"], config["start_step"], config["update_steps"]
        self.current_level = self.start_level
        self.register_buffer("mask", torch.zeros(self.n_level * self.n_features_per_level),
                             persistent=False)

    def forward(self, x):
        return self.encoding(x, n_active_levels=self.current_level)

    def update_step(self, epoch, global_step):
        self.current_level = min(
            self.start_level + max(global_step - self.start_step, 0) // self.update_steps,
            self.n_level)
        self.mask[: self.current_level * self.n_features_per_level] = 1.0


class CompositeEncoding(nn.Module):
    """cat([x * xyz_scale + xyz_offset, encoding(x)]) (:78-79), written by one kernel when the inner
    encoding is a hash grid."""

    def __init__(self, encoding, include_xyz=False, xyz_scale=1.0, xyz_offset=0.0):
        super().__init__()
        self.encoding = encoding
        self.include_xyz, self.xyz_scale, self.xyz_offset = include_xyz, xyz_scale, xyz_offset
        self.n_output_dims = int(include_xyz) * encoding.n_input_dims + encoding.n_output_dims

    def _hash(self):
        e = self.encoding
        if isinstance(e, ProgressiveBandHashGrid):
            return e.encoding, e.current_level
        if isinstance(e, tcnn.Encoding):
            return e, None
        return None, None

    def forward(self, x, *args, fd7_eps_unit=None):
        grid, n_active = self._hash()
        if grid is not None:
            return ops.hashgrid_encode(x.reshape(-1, 3), grid.params, grid.meta,
                                       n_active_levels=n_active, include_xyz=self.include_xyz,
                                       xyz_scale=self.xyz_scale, xyz_offset=self.xyz_offset,
                                       fd7_eps_unit=fd7_eps_unit)
        enc = self.encoding(x, *args)
        if not self.include_xyz:
            return enc
        return torch.cat([x * self.xyz_scale + self.xyz_offset, enc], dim=-1)

    def update_step(self, epoch, global_step):
        update_module_step(self.encoding, epoch, global_step)

    def regularizations(self):
        return self.encoding.regularizations() if hasattr(self.encoding, "regularizations") else {}


def get_encoding(n_input_dims, config):
    """Input is expected in [0,1] (:91-106)."""
    if config.otype == "ProgressiveBandHashGrid":
        encoding = ProgressiveBandHashGrid(n_input_dims, config_to_primitive(config))
    elif config.otype in ("HashGrid", "Grid"):
        encoding = tcnn.Encoding(n_input_dims, config_to_primitive(config))
    else:
        raise NotImplementedError(f"encoding otype {config.otype!r} is outside the geometry hot path")
    return CompositeEncoding(encoding, include_xyz=config.get("include_xyz", False),
                             xyz_scale=config.get("xyz_scale", 2.0),
                             xyz_offset=config.get("xyz_offset", -1.0))


class VanillaMLP(nn.Module):
    """Linear -> act -> ... -> Linear with the reference's initialisation and parameter names
    (``layers.{0,2,4}.{bias,weight_g,weight_v}`` under weight_norm, ``.weight`` otherwise), each
    layer one fused fp32-MFMA kernel."""

    def __init__(self, dim_in, dim_out, config):
        super().__init__()
        self.n_neurons, self.n_hidden_layers = config["n_neurons"], config["n_hidden_layers"]
        self.sphere_init = config.get("sphere_init", False)
        self.weight_norm = config.get("weight_norm", False)
        self.sphere_init_radius = config.get("sphere_init_radius", 0.5)
        self.inside_outside = config.get("inside_outside", False)
        # 'fp32' (default: fp32-equivalent split products, the reference's nn.Linear precision) or 'bf16' (BASELINE.json
        # configs[4] "bf16 MLP on MFMA": operands rounded once to bf16, fp32 accumulation, fp32 master weights); a
        # per-network opt-in key of the network's config node that the reference's yaml simply does not carry
        self._wn_cache = {}     # id(layer) -> ((versions, pointers), W, has_graph): see _normed_weight
        self.precision = str(config.get("precision", "fp32")).lower()
        if self.precision not in ("fp32", "bf16", "fp16"):
            raise ValueError(f"VanillaMLP precision {self.precision!r}: fp32, bf16 or fp16")
        layers = [self.make_linear(dim_in, self.n_neurons, True, False), self.make_activation()]
        for _ in range(self.n_hidden_layers - 1):
            layers += [self.make_linear(self.n_neurons, self.n_neurons, False, False),
                       self.make_activation()]
        layers += [self.make_linear(self.n_neurons, dim_out, False, True)]
        self.layers = nn.Sequential(*layers)
        self.hidden_act = "softplus100" if self.sphere_init else "relu"
        out_act = str(config.get("output_activation", "none")).lower()
        if out_act not in ("none", "sigmoid", "relu"):
            raise NotImplementedError(f"output_activation {out_act!r}")
        self.output_act = out_act
        # columns of the input that need a gradient (None = all).  The SDF network's first three
        # inputs are the xyz pass-through of a non-differentiable position under FD normals.
        self.input_grad_cols = None

    def make_linear(self, dim_in, dim_out, is_first, is_last):
        layer = nn.Linear(dim_in, dim_out, bias=True)
        if self.sphere_init:
            if is_last:
                sgn = -1.0 if self.inside_outside else 1.0
                nn.init.constant_(layer.bias, -sgn * self.sphere_init_radius)
                nn.init.normal_(layer.weight, mean=sgn * math.sqrt(math.pi) / math.sqrt(dim_in),
                                std=0.0001)
            elif is_first:
                nn.init.constant_(layer.bias, 0.0)
                nn.init.constant_(layer.weight[:, 3:], 0.0)
                nn.init.normal_(layer.weight[:, :3], 0.0, math.sqrt(2) / math.sqrt(dim_out))
            else:
                nn.init.constant_(layer.bias, 0.0)
                nn.init.normal_(layer.weight, 0.0, math.sqrt(2) / math.sqrt(dim_out))
        else:
            nn.init.constant_(layer.bias, 0.0)
            nn.init.kaiming_uniform_(layer.weight, nonlinearity="relu")
        if self.weight_norm:
            layer = nn.utils.weight_norm(layer)
        return layer

    def make_activation(self):
        return nn.Softplus(beta=100) if self.sphere_init else nn.ReLU(inplace=True)

    def __getstate__(self):
        # (copy.deepcopy / pickle of a module that has run: the cached W tensors carry autograd nodes and are not state)
        d = self.__dict__.copy()
        d["_wn_cache"] = {}
        return d

    def _normed_weight(self, m):
        """weight_norm(g, v) of one layer, computed ONCE per parameter version: a training step evaluates the SDF network
        five times (sampling, render, secondary rays, their sampling, curvature), which re-normalised every layer each time
        and ran one weight-norm backward per evaluation.  The cached W carries its autograd node, so the evaluations'
        weight gradients sum there and ONE backward runs; the cache is dropped when that backward has run (the graph is
        spent), when g or v change (their version counters), or when gradient tracking is needed but the cached W has none."""
        g, v = m.weight_g, m.weight_v
        want_graph = torch.is_grad_enabled() and (g.requires_grad or v.requires_grad)
        key = (g._version, v._version, g.data_ptr(), v.data_ptr())
        c = self._wn_cache.get(id(m))
        if c is not None and c[0] == key and (c[2] or not want_graph):
            # W was produced on the stream recorded with it: a consumer on ANOTHER stream (chunk_batch(streams=2) under
            # no_grad, the bench's two chunks in flight) waits for that event and tells the allocator the tensor is in use
            # there (the cache used to rely on an unrelated host sync between the two chunks)
            if w_is_cuda(c[1]):
                cur = torch.cuda.current_stream(c[1].device)
                if c[3] is not None and cur != c[4]:
                    cur.wait_event(c[3])
                    c[1].record_stream(cur)
            return c[1] if want_graph else c[1].detach()
        with torch.enable_grad() if (g.requires_grad or v.requires_grad) else torch.no_grad():
            w = ops.weight_norm(g, v)
        has_graph = w.requires_grad
        if has_graph:
            cache, mid = self._wn_cache, id(m)

            def spent(grad):           # (a tensor hook must return None or a tensor)
                cache.pop(mid, None)
            w.register_hook(spent)
        ev = st = None
        if w_is_cuda(w):
            st = torch.cuda.current_stream(w.device)
            ev = torch.cuda.Event()
            ev.record(st)
        self._wn_cache[id(m)] = (key, w, has_graph, ev, st)
        return w if want_graph else w.detach()

    def invalidate_weight_cache(self):
        """Writes that bypass the version counters (``p.data.copy_()``, ``p.data.mul_()``) leave a stale W behind: call this
        after them.  ``load_state_dict`` and ``train()`` / ``eval()`` switches do it themselves."""
        self._wn_cache.clear()

    def _load_from_state_dict(self, *a, **kw):      # (load_state_dict of this module or of any parent comes through here)
        self.__dict__.get("_wn_cache", {}).clear()
        return super()._load_from_state_dict(*a, **kw)

    def train(self, mode=True):
        self.__dict__.get("_wn_cache", {}).clear()
        return super().train(mode)

    def effective_weights(self):
        """[(W [out,in], b [out])] per Linear; W = weight_norm(g, v) through the HIP kernel."""
        out = []
        for m in self.layers:
            if isinstance(m, nn.Linear):
                w = self._normed_weight(m) if self.weight_norm else m.weight
                out.append((w, m.bias))
        return out

    def forward(self, x, out_act=None, x2=None):
        """``out_act`` overrides the configured output activation (lets a caller fuse e.g. the texture
        networks' color_activation into the last layer's kernel).  ``x2``: the input is cat([x, x2], -1) (the radiance
        networks' [feature, encoding], models/texture.py:299-313) without the caller having to materialise it."""
        wb = self.effective_weights()
        acts = [self.hidden_act] * (len(wb) - 1) + [out_act or self.output_act]
        return ops.mlp_chain(x.float(), wb, acts, dx_cols=self.input_grad_cols, precision=self.precision,
                             x2=None if x2 is None else x2.float())


def w_is_cuda(t):
    return t is not None and t.is_cuda


def get_mlp(n_input_dims, n_output_dims, config):
    if config.otype == "VanillaMLP":
        return VanillaMLP(n_input_dims, n_output_dims, config_to_primitive(config))
    if config.otype == "Identity":
        return nn.Identity()
    raise NotImplementedError(f"network otype {config.otype!r}: the shipped configs use VanillaMLP "
                              "(configs/split-mixed-occ-tensoir.yaml:75)")


class EncodingWithNetwork(nn.Module):
    def __init__(self, encoding, network):
        super().__init__()
        self.encoding, self.network = encoding, network

    def forward(self, x):
        return self.network(self.encoding(x))

    def update_step(self, epoch, global_step):
        update_module_step(self.encoding, epoch, global_step)
        update_module_step(self.network, epoch, global_step)


def get_encoding_with_network(n_input_dims, n_output_dims, encoding_config, network_config):
    encoding = get_encoding(n_input_dims, encoding_config)
    network = get_mlp(encoding.n_output_dims, n_output_dims, network_config)
    return EncodingWithNetwork(encoding, network)
