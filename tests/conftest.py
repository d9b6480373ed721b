import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


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
