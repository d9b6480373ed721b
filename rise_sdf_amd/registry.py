"""Name -> class registry with the interface of the reference's ``models/__init__.py:1-14`` (``models`` dict,
``@register(name)``, ``make(name, config)``), so that configs keep addressing modules by the same names
(``volume-sdf``, ``neus``, ``split-mixed-occ``, ``volume-mixed-mip-split-occ``, ``volume-radiance``,
``envlight-mip-cube``)."""
from __future__ import annotations

from typing import Callable, Dict, Type

models: Dict[str, Type] = {}


def register(name: str) -> Callable[[Type], Type]:
    """Class decorator: file ``cls`` under ``name``.  Re-registering a name replaces the entry (the reference's
    behaviour; INTEGRATION.md uses it to swap implementations)."""
    def _file(cls: Type) -> Type:
        models[name] = cls
        return cls
    return _file


def make(name: str, config):
    """Instantiate the class registered under ``name`` with its config node."""
    try:
        cls = models[name]
    except KeyError:
        raise KeyError(f"no model registered as {name!r}; known: {sorted(models)}") from None
    return cls(config)
