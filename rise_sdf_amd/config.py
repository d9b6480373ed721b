"""Minimal OmegaConf-node stand-in.

The reference passes OmegaConf nodes to every constructor (``models.make(name, config)``,
models/base.py:7-14) and reads them with attribute access, ``.get(key, default)`` and ``in``.
OmegaConf is not available in this image; ``Config`` gives plain nested dicts the same read
interface (a real OmegaConf node also works wherever a Config is accepted).
"""
from __future__ import annotations


class Config(dict):
    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return Config(v) if isinstance(v, dict) and not isinstance(v, Config) else v

    def __setattr__(self, k, v):
        self[k] = v

    def get(self, k, default=None):
        """OmegaConf's ``node.get`` returns sub-nodes, not plain dicts."""
        v = dict.get(self, k, default)
        return Config(v) if isinstance(v, dict) and not isinstance(v, Config) else v

    def copy(self):
        return Config(dict.copy(self))


def config_to_primitive(config):
    """utils/misc.py config_to_primitive: OmegaConf node -> plain dict."""
    if isinstance(config, dict):
        return {k: config_to_primitive(v) for k, v in config.items()}
    try:  # a genuine OmegaConf node, when omegaconf is installed
        from omegaconf import OmegaConf
        return OmegaConf.to_container(config, resolve=True)
    except Exception:
        return config


def load_yaml(path, overrides=None):
    """Read a reference YAML (configs/*.yaml) without OmegaConf: resolves ${a.b} references and the
    custom resolvers of utils/misc.py:6-13 (add, sub, mul, div, idiv, basename,
    calc_exp_lr_decay_rate)."""
    import os
    import re

    import yaml

    with open(path) as f:
        root = yaml.safe_load(f)
    for k, v in (overrides or {}).items():
        node = root
        parts = k.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = v

    fns = {"add": lambda a, b: a + b, "sub": lambda a, b: a - b, "mul": lambda a, b: a * b,
           "div": lambda a, b: a / b, "idiv": lambda a, b: a // b,
           "basename": lambda p: os.path.basename(p),
           "calc_exp_lr_decay_rate": lambda factor, n: factor ** (1.0 / n)}
    pat = re.compile(r"\$\{([^${}]+)\}")

    def lookup(ref):
        node = root
        for p in ref.split("."):
            node = node[p]
        return resolve(node)

    def num(s):
        if not isinstance(s, str):
            return s
        try:
            return int(s)
        except ValueError:
            try:
                return float(s)
            except ValueError:
                return s

    def resolve(v):
        if isinstance(v, dict):
            return {k: resolve(x) for k, x in v.items()}
        if isinstance(v, list):
            return [resolve(x) for x in v]
        if not isinstance(v, str):
            return v
        while True:
            m = pat.search(v)
            if not m:
                return num(v) if v != "" else v
            expr = m.group(1)
            if ":" in expr:
                name, args = expr.split(":", 1)
                val = fns[name](*[num(a.strip()) for a in args.split(",")])
            else:
                val = lookup(expr)
            if m.start() == 0 and m.end() == len(v):
                return val
            v = v[:m.start()] + str(val) + v[m.end():]

    def deep(v):
        # innermost-first resolution: repeat until stable
        prev = None
        while prev != v:
            prev, v = v, resolve(v)
        return v

    return Config(deep(root))


def tensoir_model_config(n_levels=16, log2_T=19, hidden=128, tex_precision="fp32", sdf_precision="fp32", **overrides):
    """The ``model:`` node of configs/split-mixed-occ-tensoir.yaml:31-133 as a Config (sizes, schedules and flags of the
    shipped yaml: 128-wide SDF and radiance MLPs, 48 features, 16-level base-32 T = 2^19 grid, 512^2 environment cube,
    occupancy pruning, secondary rays, split-sum from step 10000).  ``tex_precision`` / ``sdf_precision``: this build's
    per-network ``precision`` key (fp32 or bf16; BASELINE.json configs[4]).  ``overrides`` replace top-level keys."""
    mlp = lambda n: {"otype": "VanillaMLP", "activation": "ReLU", "output_activation": "none", "n_neurons": hidden,   # noqa: E731
                     "n_hidden_layers": n, "precision": tex_precision}
    cfg = {
        "name": "split-mixed-occ", "indirect_pred": True, "relighting_threshold": 0.3, "radius": 1.5,
        "num_samples_per_ray": 1024, "num_samples_per_secondary_ray": 96, "train_num_rays": 256,
        "max_train_num_rays": 4096, "grid_prune": True, "grid_prune_occ_thre": 0.001, "dynamic_ray_sampling": True,
        "randomized": True, "ray_chunk": 4096, "cos_anneal_end": 10000, "learned_background": False,
        "split_sum_kick_in_step": 10000, "background_color": "random",
        "variance": {"init_val": 0.3, "modulate": False},
        "geometry": {
            "name": "volume-sdf", "radius": 1.5, "feature_dim": 48, "grad_type": "finite_difference",
            "finite_difference_eps": "progressive",
            "xyz_encoding_config": {"otype": "ProgressiveBandHashGrid", "n_levels": n_levels, "start_level": 6,
                                    "start_step": 6000, "update_steps": 500, "n_features_per_level": 2,
                                    "log2_hashmap_size": log2_T, "base_resolution": 32,
                                    "per_level_scale": 1.447269237440378, "include_xyz": True},
            "mlp_network_config": {"otype": "VanillaMLP", "activation": "ReLU", "output_activation": "none",
                                   "n_neurons": hidden, "n_hidden_layers": 2, "sphere_init": True,
                                   "sphere_init_radius": 0.5, "weight_norm": True, "precision": sdf_precision}},
        "texture": {"name": "volume-mixed-mip-split-occ", "input_feature_dim": 48, "other_dim": 3, "sample_size": 8,
                    "dir_encoding_config": {"otype": "SphericalHarmonics", "degree": 5, "reflected": True},
                    "metallic_mlp_network_config": mlp(2), "albedo_mlp_network_config": mlp(4),
                    "spec_mlp_network_config": mlp(4), "roughness_mlp_network_config": mlp(2),
                    "secondary_mlp_network_config": mlp(4),
                    "xyz_encoding_config": {"otype": "VanillaFrequency", "n_frequencies": 6},
                    "color_activation": "sigmoid"},
        "light": {"name": "envlight-mip-cube",
                  "envlight_config": {"hdr_filepath": None, "clamp": True, "nmf_format": False, "scale": 0.5,
                                      "bias": 0.25, "base_res": 512}},
    }
    cfg.update(overrides)
    return Config(cfg)


# system.loss of the same yaml (:140-152) and the per-module Adam learning rates (:153-167)
TENSOIR_LAMBDAS = {"lambda_rgb_mse": 10.0, "lambda_rgb_l1": 0.0, "lambda_rgb_phys_mse": 10.0, "lambda_rgb_phys_l1": 0.0,
                   "lambda_mask": 0.1, "lambda_eikonal": 0.05, "lambda_sparsity": 0.01, "lambda_curvature": 1.0,
                   "lambda_opaque": 0.0}
TENSOIR_REG_LAMBDAS = {"lambda_normal_orientation": 0.05}
TENSOIR_LRS = {"geometry": 0.005, "texture": 0.005, "variance": 0.001, "emitter": 0.01}


def tensoir_optimizer(model, lrs=None, fused=False):
    """systems/utils.py:314-346 for the yaml's optimizer node: Adam(betas (0.9, 0.999), eps 1e-12), one parameter group
    per sub-module with its own learning rate.  ``fused``: torch's fused multi-tensor Adam (one kernel per group instead
    of ~9 foreach passes over every parameter: the same update rule; opt-in, the reference's parse_optimizer passes the
    yaml's args only)."""
    import torch
    lrs = dict(TENSOIR_LRS if lrs is None else lrs)
    groups = [{"params": list(getattr(model, k).parameters()), "lr": lr} for k, lr in lrs.items()
              if getattr(model, k, None) is not None]
    kw = {"fused": True} if fused else {}
    return torch.optim.Adam(groups, lr=0.005, betas=(0.9, 0.999), eps=1e-12, **kw)
