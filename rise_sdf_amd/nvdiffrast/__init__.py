"""Drop-in for the one nvdiffrast entry RISE-SDF's hot path calls: ``import nvdiffrast.torch as dr`` ->
``dr.texture`` (models/texture.py:340, lib/pbr/light.py:194-206,259-262, lib/pbr/utils/light_utils.py:108)."""
from . import torch  # noqa: F401
