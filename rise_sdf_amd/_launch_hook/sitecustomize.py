"""Child ranks of the zero-edit launcher (rise_sdf_amd/launch.py).

With ``--gpu 0,1`` the reference builds ``Trainer(strategy='ddp')`` (launch.py:84-97) and Lightning's subprocess launcher
re-creates ranks 1..N-1 as ``[sys.executable, sys.argv[0]] + sys.argv[1:]`` -- i.e. ``python launch.py ...``, which would
bypass the drop-ins and the registry swap that ``python -m rise_sdf_amd.launch`` put in place for rank 0.  The launcher
therefore exports ``RSDF_LAUNCH_SCRIPT`` / ``RSDF_LAUNCH_FUSED`` and puts this directory first on ``PYTHONPATH``: every
Python interpreter started from it imports this ``sitecustomize`` at start-up, which runs the same ``prepare()`` when (and
only when) the interpreter is about to execute that very script.  A ``sitecustomize`` that this one shadows is imported
afterwards."""
import os
import sys


def _chain():
    here = os.path.dirname(os.path.abspath(__file__))
    rest = [p for p in sys.path if os.path.abspath(p or ".") != here]
    import importlib.machinery
    spec = importlib.machinery.PathFinder.find_spec("sitecustomize", rest)
    if spec is not None and spec.loader is not None:
        import importlib.util
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)


def _prepare():
    script = os.environ.get("RSDF_LAUNCH_SCRIPT")
    if not script:
        return
    argv = getattr(sys, "orig_argv", None) or getattr(sys, "argv", [])
    # only the interpreter that runs the script itself (python [opts] <script> ...), not every helper process
    if not any(os.path.abspath(a) == script for a in argv[1:3] if not a.startswith("-")):
        return
    if "rise_sdf_amd.launch" in " ".join(argv[:4]):       # rank 0 goes through launch.main() itself
        return
    pkg_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if pkg_root not in sys.path:
        sys.path.insert(1, pkg_root)
    from rise_sdf_amd import launch
    launch.prepare(script, os.environ.get("RSDF_LAUNCH_FUSED", "1") != "0")


try:
    _prepare()
finally:
    _chain()
