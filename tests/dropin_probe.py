#!/usr/bin/env python3
"""Build-container probe behind tests/test_dropin_reference.py (run as a subprocess so the reference's top-level
packages ``models``/``systems``/``lib``/``utils`` never enter the pytest process).

1. ``rise_sdf_amd.dropin.install()`` fills nerfacc / nerfacc.volrend / tinycudann / nvdiffrast.torch /
   lib.nerfacc.cuda._backend; packages that are absent from this image and NOT on the hot path
   (pytorch_lightning, omegaconf, imageio, ...) get permissive stubs, exactly as tests/golden/make_golden.py does.
2. The reference modules that bind to those surfaces are imported from /root/reference (read-only, never copied).
3. Every call site of a drop-in surface in those modules is found by AST and its positional count / keyword names
   are bound against the mirror's signature with ``inspect.signature(...).bind``.
Prints one JSON object: {"imported": [...], "checked": n, "failures": [...]}.
"""
import ast
import importlib
import inspect
import json
import os
import sys

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, "golden"))


def main():
    import make_golden as mg                       # stub machinery only (no reference code is executed by import)
    mg.install_stubs()                             # absent off-path packages; also puts REF on sys.path
    import rise_sdf_amd.dropin as dropin
    dropin.install(patch_renderutils=False)        # ... and the hot-path surfaces are replaced by the HIP mirrors
    for slot in dropin.SLOTS:
        assert sys.modules[slot].__name__.startswith("rise_sdf_amd."), slot
    os.chdir(REF)

    modules = ["models.neus", "models.volrend", "models.split_mixed_occ", "models.texture", "models.network_utils",
               "models.geometry", "lib.pbr.light", "lib.pbr.utils.light_utils", "lib.nerfacc", "lib.renderutils.ops"]
    imported = []
    for m in modules:
        importlib.import_module(m)
        imported.append(m)
    dropin.install(patch_renderutils=True)
    import lib.renderutils.ops as ru_ops
    import rise_sdf_amd.renderutils as R
    assert ru_ops._get_plugin() is R.plugin

    # B1: the registry swap of INTEGRATION.md -- the reference's own ``models.make`` then builds this repo's classes
    import models as ref_models
    import rise_sdf_amd
    for name in ("volume-sdf", "neus", "split-mixed-occ", "volume-mixed-mip-split-occ", "volume-radiance",
                 "envlight-mip-cube"):
        assert name in ref_models.models, name                     # the reference registered it under this name
        ref_models.models[name] = rise_sdf_amd.models[name]
    cfg = rise_sdf_amd.Config({"name": "volume-radiance", "input_feature_dim": 16,
                               "dir_encoding_config": {"otype": "SphericalHarmonics", "degree": 4},
                               "mlp_network_config": {"otype": "VanillaMLP", "activation": "ReLU",
                                                      "output_activation": "none", "n_neurons": 16, "n_hidden_layers": 1},
                               "color_activation": "sigmoid"})
    built = ref_models.make("volume-radiance", cfg)
    assert type(built).__module__.startswith("rise_sdf_amd."), type(built)

    import nerfacc
    import nvdiffrast.torch as dr
    import tinycudann as tcnn
    from rise_sdf_amd import renderutils as ru
    from rise_sdf_amd.nerfacc import cuda as C

    # surface name -> callable whose signature must accept the reference's call
    by_name = {
        "render_weight_from_density": nerfacc.render_weight_from_density,
        "render_weight_from_alpha": nerfacc.render_weight_from_alpha,
        "accumulate_along_rays": nerfacc.accumulate_along_rays,
        "ray_aabb_intersect": nerfacc.ray_aabb_intersect,
        "OccGridEstimator": nerfacc.OccGridEstimator,
    }
    by_attr = {
        ("dr", "texture"): dr.texture,
        ("ru", "diffuse_cubemap"): ru.diffuse_cubemap,
        ("ru", "specular_cubemap"): ru.specular_cubemap,
        ("tcnn", "Encoding"): tcnn.Encoding.__init__,
        ("tcnn", "free_temporary_memory"): tcnn.free_temporary_memory,
    }
    by_method = {                                  # estimator methods, whatever the receiver expression is
        "sampling": nerfacc.OccGridEstimator.sampling,
        "update_every_n_steps": nerfacc.OccGridEstimator.update_every_n_steps,
    }
    plugin_methods = {n: getattr(R.plugin, n) for n in
                      ("diffuse_cubemap_fwd", "diffuse_cubemap_bwd", "specular_bounds", "specular_cubemap_fwd",
                       "specular_cubemap_bwd")}
    c_funcs = {n: getattr(C, n) for n in dir(C) if not n.startswith("_") and callable(getattr(C, n))}

    files = ["models/neus.py", "models/volrend.py", "models/split_mixed_occ.py", "models/texture.py",
             "models/network_utils.py", "models/utils.py", "lib/pbr/light.py", "lib/pbr/utils/light_utils.py",
             "lib/renderutils/ops.py", "lib/nerfacc/ray_marching.py", "lib/nerfacc/vol_rendering.py",
             "lib/nerfacc/grid.py", "lib/nerfacc/pack.py", "lib/nerfacc/intersection.py",
             "lib/nerfacc/contraction.py"]
    checked, failures = 0, []

    def try_bind(fn, call, where, self_arg=False):
        nonlocal checked
        if any(isinstance(a, ast.Starred) for a in call.args) or any(k.arg is None for k in call.keywords):
            return
        args = [None] * (len(call.args) + (1 if self_arg else 0))
        kwargs = {k.arg: None for k in call.keywords}
        checked += 1
        try:
            inspect.signature(fn).bind(*args, **kwargs)
        except TypeError as e:
            failures.append(f"{where}: {e}")

    for rel in files:
        tree = ast.parse(open(os.path.join(REF, rel)).read())
        # names this file pulls from the vendored package would shadow the 0.5.3 ones: honour the import source
        vendored = set()
        for node in ast.walk(tree):
            if isinstance(node, ast.ImportFrom) and node.module and (node.module.startswith("lib.nerfacc") or
                                                                     rel.startswith("lib/nerfacc") and node.level > 0):
                vendored.update(a.asname or a.name for a in node.names)
        for node in ast.walk(tree):
            if not isinstance(node, ast.Call):
                continue
            where = f"{rel}:{node.lineno}"
            f = node.func
            if isinstance(f, ast.Name) and f.id in by_name and f.id not in vendored and not rel.startswith("lib/nerfacc"):
                fn = by_name[f.id]
                try_bind(fn.__init__ if inspect.isclass(fn) else fn, node, where, self_arg=inspect.isclass(fn))
            elif isinstance(f, ast.Attribute) and isinstance(f.value, ast.Name) and (f.value.id, f.attr) in by_attr:
                fn = by_attr[(f.value.id, f.attr)]
                try_bind(fn, node, where, self_arg=(f.attr == "Encoding"))
            elif isinstance(f, ast.Attribute) and f.attr in by_method and not rel.startswith("lib/"):
                try_bind(by_method[f.attr], node, where, self_arg=True)
            elif isinstance(f, ast.Attribute) and f.attr in plugin_methods and rel == "lib/renderutils/ops.py":
                try_bind(plugin_methods[f.attr], node, where)
            elif (isinstance(f, ast.Attribute) and isinstance(f.value, ast.Name) and f.value.id == "_C"
                  and rel.startswith("lib/nerfacc")):
                name = "ContractionType" if f.attr == "ContractionTypeGetter" else f.attr
                if name not in c_funcs and name != "ContractionType":
                    failures.append(f"{where}: _C.{f.attr} missing from rise_sdf_amd.nerfacc.cuda")
                    continue
                checked += 1
                if name != "ContractionType":
                    try_bind(c_funcs[name], node, where)
                    checked -= 1
    print(json.dumps({"imported": imported, "checked": checked, "failures": failures}))


if __name__ == "__main__":
    main()
