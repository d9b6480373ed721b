"""RCCL executes: the gradient exchange of the N-rank step (launch.py:84-97) on a one-rank ``nccl``
group on cuda:0, in a fresh child process (the suite's own process must not own a process group).  Multi-rank VALUES are
covered over gloo (tests/test_dist_gloo.py, tests/test_gpu_dist_step.py); this is the backend the multi-GPU bench uses."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_one_rank_group_runs_the_gradient_exchange(dev):
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))                    # a free rendezvous port (the probe's default may be taken on a shared box)
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT=str(port))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RSDF_DIST_SHARE_GPU", "RSDF_DIST_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_probe.py")], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert res["dist_backend"] == "nccl" and res["world"] == 1, res
    # the table (14.5 M fp32) and the 6 x 512 x 512 x 3 environment map go in place, everything else through the flat buffer
    assert 14533536 in res["in_place_tensors"] and 6 * 512 * 512 * 3 in res["in_place_tensors"], res
    assert res["flat_elements"] > 0 and res["collectives_per_step"] == len(res["in_place_tensors"]) + 1, res
    assert res["bit_identical"] and res["table_sum_identity"], res
