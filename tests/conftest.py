import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    if os.environ.get("RSDF_GUARD_ALLOC") == "1":
        # every device allocation of this run ends flush against an unmapped page (tests/guard_alloc.cpp): an access past
        # the end of ANY tensor faults on the spot instead of landing in the caching allocator's neighbouring block
        import torch
        so = os.path.join(ROOT, "tests", "_guard_alloc.so")
        if not os.path.exists(so):      # normally built by __graft_entry__.build()
            import subprocess
            subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", "-shared", "-fPIC", "-w",
                                   os.path.join(ROOT, "tests", "guard_alloc.cpp"), "-o", so])
        torch.cuda.memory.change_current_allocator(
            torch.cuda.memory.CUDAPluggableAllocator(so, "guard_malloc", "guard_free"))


def _trace_backward(path):
    """RSDF_TRACE_BACKWARD=<file> (debug aid for GPU memory faults inside a backward pass, which abort the process from a
    runtime thread): every autograd node of every ``Tensor.backward()`` call is named in <file> and the device drained BEFORE it
    runs, so the last line of the file is the node whose kernels faulted (rsdf entry points: RSDF_DEBUG_SYNC)."""
    import torch
    orig = torch.Tensor.backward

    def pre(name):
        def hook(_grads):
            torch.cuda.synchronize()
            with open(path, "a") as f:
                f.write(name + "\n")
        return hook

    def backward(self, *a, **kw):
        seen, stack = set(), [self.grad_fn]
        while stack:
            n = stack.pop()
            if n is None or n in seen:
                continue
            seen.add(n)
            n.register_prehook(pre(n.name()))
            stack += [fn for fn, _ in n.next_functions]
        with open(path, "a") as f:
            f.write(f"-- backward over {len(seen)} nodes\n")
        return orig(self, *a, **kw)

    torch.Tensor.backward = backward


if os.environ.get("RSDF_TRACE_BACKWARD"):
    _trace_backward(os.environ["RSDF_TRACE_BACKWARD"])


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _fresh_status_words(request):
    """The kernels' sticky status words (rise_sdf_amd._lib.poll_status) belong to the test that set them: a test that
    drives the x2 field out of its range on purpose must not make the next test's marcher raise."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        from rise_sdf_amd import _lib
        if _lib._STATUS:
            _lib.poll_status(raise_on_error=False)
        _lib.reset_range_free()          # (a test that tripped the forward range guard must not switch the next test's kernels)
