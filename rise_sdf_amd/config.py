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
