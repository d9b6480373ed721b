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
    argv = list(getattr(sys, "orig_argv", None) or getattr(sys, "argv", []))
    # only the interpreter that runs the script itself (python [opts] <script> ...), not every helper process: the FIRST
    # non-option argument, skipping the values of the options that take one (python -u -X faulthandler -W ignore launch.py)
    first, i = None, 1
    while i < len(argv):
        a = argv[i]
        if a in ("-c", "-m"):             # python -c ... / -m module: not a script run (rank 0 is `-m rise_sdf_amd.launch`)
            return
        if a in ("-W", "-X", "--check-hash-based-pycs"):
            i += 2
            continue
        if a.startswith("-") and a != "-":
            i += 1
            continue
        first = a
        break
    if first is None or os.path.abspath(first) != script:
        return
    pkg_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if pkg_root not in sys.path:
        sys.path.insert(1, pkg_root)
    from rise_sdf_amd import launch
    launch.prepare(script, os.environ.get("RSDF_LAUNCH_FUSED", "1") != "0")
    # this interpreter is set up; processes IT starts that are not Lightning ranks of the same script (dataloader workers fork
    # and inherit the modules; tools the script spawns start clean) need neither the hook nor the variables.  Lightning's rank
    # children are created from os.environ as it is when the Trainer launches them, so the launcher's variables are kept
    # while LOCAL_RANK is unset (this is a rank that may still spawn its siblings) and dropped in a spawned rank.
    if os.environ.get("LOCAL_RANK") not in (None, "0"):
        here = os.path.dirname(os.path.abspath(__file__))
        pp = [q for q in os.environ.get("PYTHONPATH", "").split(os.pathsep) if q and os.path.abspath(q) != here]
        if pp:
            os.environ["PYTHONPATH"] = os.pathsep.join(pp)
        else:
            os.environ.pop("PYTHONPATH", None)
        os.environ.pop("RSDF_LAUNCH_SCRIPT", None)
        os.environ.pop("RSDF_LAUNCH_FUSED", None)


try:
    _prepare()
finally:
    _chain()
