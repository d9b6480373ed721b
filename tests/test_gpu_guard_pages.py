"""Entry points on buffers that end flush against an unmapped page: an access past the end of an operand is a GPU memory
fault, which aborts the process -- so each case (tests/guard_page_cases.py) runs in a child process and a fault is its
non-zero exit code.  (`RSDF_GUARD_ALLOC=1 pytest -m gpu` puts the WHOLE suite on such buffers: tests/guard_alloc.cpp.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def _case_names():
    # the names only (the module imports torch and the library; listing must work without a GPU)
    import re
    src = open(os.path.join(HERE, "guard_page_cases.py")).read()
    return re.findall(r'^    "(\w+)": (?:lambda|\w+,)', src, flags=re.M)


@pytest.mark.parametrize("case", _case_names())
def test_no_access_past_the_end(case, dev):
    r = subprocess.run([sys.executable, os.path.join(HERE, "guard_page_cases.py"), case], capture_output=True, text=True,
                       timeout=600)
    tail = "\n".join((r.stdout + r.stderr).splitlines()[-6:])
    assert r.returncode == 0, f"{case}: exit code {r.returncode} (134 = GPU memory fault); last lines:\n{tail}"
    assert r.stdout.rstrip().endswith(f"{case}: ok")
