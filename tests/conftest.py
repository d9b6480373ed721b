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
        assert os.path.exists(so), "RSDF_GUARD_ALLOC=1 needs tests/_guard_alloc.so (__graft_entry__.build())"
        torch.cuda.memory.change_current_allocator(
            torch.cuda.memory.CUDAPluggableAllocator(so, "guard_malloc", "guard_free"))


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
