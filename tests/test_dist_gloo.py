"""CPU, world_size 2 over gloo: the ray-parallel sharding helpers and the gradient all-reduce
(the only exchange step of the path) behave like DDP's mean all-reduce."""
import os
import socket
import subprocess
import sys
import textwrap

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


WORKER = textwrap.dedent("""
    import os, sys, torch
    sys.path.insert(0, os.environ["RSDF_ROOT"])
    from rise_sdf_amd import dist as rd
    rank, local, world = rd.init_from_env(backend="gloo")
    assert world == 2
    torch.manual_seed(rd.rank_seed(0, rank))
    big = torch.nn.Parameter(torch.zeros(1000))
    small = [torch.nn.Parameter(torch.zeros(7, 3)), torch.nn.Parameter(torch.zeros(()))]
    big.grad = torch.full((1000,), float(rank + 1))
    small[0].grad = torch.full((7, 3), 10.0 * (rank + 1))
    small[1].grad = None if rank == 0 else torch.tensor(4.0)     # a parameter unused on one rank
    buckets = rd.GradBuckets([big] + small)
    buckets.all_reduce_mean(world)
    assert torch.allclose(big.grad, torch.full((1000,), 1.5))
    assert torch.allclose(small[0].grad, torch.full((7, 3), 15.0))
    assert torch.allclose(small[1].grad, torch.tensor(2.0))
    # several large parameters (hash table + environment map) are reduced in place, the rest through the flat buffer;
    # async issue + finish (what TrainStep does around its host-side bookkeeping)
    rd.GradBuckets.IN_PLACE_MIN = 500
    table, envmap, w = (torch.nn.Parameter(torch.zeros(1000)), torch.nn.Parameter(torch.zeros(6, 10, 10, 3)),
                        torch.nn.Parameter(torch.zeros(4, 4)))
    table.grad, envmap.grad, w.grad = (torch.full((1000,), 2.0 * rank), torch.full((6, 10, 10, 3), 1.0 + rank),
                                       torch.full((4, 4), 3.0 - rank))
    b2 = rd.GradBuckets([table, envmap, w])
    assert len(b2.big) == 2 and len(b2.rest) == 1 and b2.flat.numel() == 16 and b2.bytes_per_step() == 4 * (1000 + 1800 + 16)
    ptrs = (table.grad.data_ptr(), envmap.grad.data_ptr())
    handles = b2.all_reduce_mean(world, async_op=True)
    assert len(handles) == 3
    b2.finish(handles, world)
    assert (table.grad.data_ptr(), envmap.grad.data_ptr()) == ptrs            # in place
    assert torch.allclose(table.grad, torch.full((1000,), 1.0)) and torch.allclose(envmap.grad, torch.full((6, 10, 10, 3), 1.5))
    assert torch.allclose(w.grad, torch.full((4, 4), 2.5))
    # chunk sharding: every chunk exactly once, round-robin
    mine = rd.shard_chunks(640000, 4096, rank, world)
    gathered = [None, None]
    torch.distributed.all_gather_object(gathered, mine)
    flat = sorted(gathered[0] + gathered[1])
    assert flat[0][0] == 0 and flat[-1][1] == 640000 and all(a[1] == b[0] for a, b in zip(flat, flat[1:]))
    assert len(gathered[0]) - len(gathered[1]) in (0, 1)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
    print("rank", rank, "ok")
""")


def test_gloo_world2_grad_allreduce(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), RSDF_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    for p in procs:
        out, _ = p.communicate(timeout=240)
        assert p.returncode == 0, out
        assert "ok" in out


def test_single_process_is_a_noop():
    from rise_sdf_amd import dist as rd
    assert rd.shard_chunks(10, 4, 0, 1) == [(0, 4), (4, 8), (8, 10)]
    p = torch.nn.Parameter(torch.zeros(3))
    p.grad = torch.ones(3)
    rd.GradBuckets([p]).all_reduce_mean(1)
    assert torch.equal(p.grad, torch.ones(3))


LAUNCHED = textwrap.dedent("""
    import json, os, sys, torch
    sys.path.insert(0, os.environ["RSDF_ROOT"])
    from rise_sdf_amd import dist as rd
    rank, local, world = rd.init_from_env(backend="gloo")
    t = torch.tensor([float(rank + 1)])
    torch.distributed.all_reduce(t)
    if len(sys.argv) > 1 and sys.argv[1] == "fail" and rank == 1:
        sys.exit(3)
    if rank == 0:
        print(json.dumps({"rccl_ranks": torch.distributed.get_world_size(), "sum": float(t)}))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
""")


def test_spawn_ranks_self_launch(tmp_path, capfd):
    """bench.py --gpus N without torchrun: N children, rank 0's line passes through, failures propagate."""
    from rise_sdf_amd import dist as rd
    script = tmp_path / "launched.py"
    script.write_text(LAUNCHED)
    os.environ["RSDF_ROOT"] = ROOT
    try:
        assert rd.spawn_ranks([sys.executable, str(script)], 2, timeout=240) == 0
        out = capfd.readouterr().out
        assert '"rccl_ranks": 2' in out and '"sum": 3.0' in out
        assert rd.spawn_ranks([sys.executable, str(script), "fail"], 2, timeout=240) == 3
    finally:
        os.environ.pop("RSDF_ROOT", None)
