"""config[3] as an N-rank training step: two ranks, each with its own ray batch through the
occupancy-pruned sampler, gradient mean through GradBuckets, identical replicas afterwards.  RCCL refuses two ranks on one
device, so on the one-GPU box the collectives run over gloo (RSDF_DIST_SHARE_GPU=1); the code path -- TrainStep,
GradBuckets on device tensors, the replicated occupancy update -- is the one ``bench.py --workload c3 --gpus N`` runs
over RCCL on a multi-GPU node."""
import json
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("stage", [0, 1])
def test_two_rank_training_step(dev, stage):
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); from rise_sdf_amd.dist import spawn_ranks; "
            "sys.exit(spawn_ranks([sys.executable, %r], 2, timeout=1500))") % (ROOT, os.path.join(ROOT, "tests", "dist_step_probe.py"))
    env = dict(os.environ, RSDF_DIST_SHARE_GPU="1", PROBE_STAGE=str(stage))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert res["stage"] == stage and res["ranks_drew_different_rays"]
    # float atomics in the scatter kernels make a replay differ in the last bits only
    assert res["replay_of_own_batch_rel"] < 1e-4, res
    assert res["allreduce_vs_single_process_rel"] < 1e-4, res
    assert res["replicas_bit_identical"], res
    assert res["occupied_cells"] > 0 and min(res["samples_rank0"]) > 0, res
