"""Ray-parallel multi-GPU support: one process per GPU, parameters replicated, rays sharded.

The reference's only parallelism is PyTorch-Lightning DDP over rays (launch.py:84-97,
``strategy='ddp'``): every rank draws its own ray batch, parameters (~74 MiB) are replicated and all
parameter gradients are mean-all-reduced once per backward.  Rays never interact, so the data path
needs no collective; the single exchange step is the gradient all-reduce, done here over RCCL
(``backend='nccl'`` on ROCm) / xGMI with two flat buffers: the hash-table gradient (55.4 MiB at the
yaml sizes) and everything else (MLPs, variance; < 1 MiB).  Ring all-reduce at p=8 moves
2(p-1)/p x 55 MiB = 97 MiB per GPU -- under 1 ms on one ~153 GB/s xGMI link, far below a step.

Known reference quirk: ``pl.seed_everything`` gives every rank the same seed, so
ranks draw identical rays.  Here rank r seeds ``seed + r``.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun).
    Returns (rank, local_rank, world_size); a no-op (0, 0, 1) when WORLD_SIZE is unset or 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # RSDF_DIST_SHARE_GPU=1 (test knob): every rank uses device 0 and the collectives go over gloo (RCCL refuses two
    # ranks on one device).  It lets the N > 1 code path -- barriers, max/sum reductions of the timing, the gradient
    # buckets on device tensors -- run on a one-GPU box; it is not a performance configuration.
    if os.environ.get("RSDF_DIST_SHARE_GPU") == "1":
        local, backend = 0, "gloo"
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("RSDF_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def rank_seed(seed: int, rank: int) -> int:
    return seed + rank


def shard_chunks(n_items: int, chunk: int, rank: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous [start, end) chunks of ``n_items`` dealt round-robin over ranks (full-image
    rendering): chunk c belongs to rank c % world."""
    out = []
    for c, s in enumerate(range(0, n_items, chunk)):
        if c % world == rank:
            out.append((s, min(s + chunk, n_items)))
    return out


class GradBuckets:
    """DDP-style gradient mean (launch.py:84-97: Lightning's DDP strategy).  Parameters of at least ``IN_PLACE_MIN``
    elements (the 55 MiB hash table, the 18 MiB environment map) are reduced IN PLACE in their ``.grad``; all remaining
    gradients (a few hundred KiB of MLP weights) are packed into one persistent flat buffer per step.  ``all_reduce_mean``
    with ``async_op=True`` only issues the collectives (RCCL runs them on its own stream); ``finish`` waits, divides and
    unpacks -- rise_sdf_amd.step.TrainStep issues right after backward and finishes just before the optimizer step."""
    IN_PLACE_MIN = 1 << 18

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        self.big = [p for p in self.params if p.numel() >= self.IN_PLACE_MIN]
        if not self.big and self.params:
            self.big = [max(self.params, key=lambda p: p.numel())]
        ids = {id(p) for p in self.big}
        self.rest = [p for p in self.params if id(p) not in ids]
        n = sum(p.numel() for p in self.rest)
        dev = self.params[0].device if self.params else "cpu"
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)

    def bytes_per_step(self) -> int:
        """Payload of one gradient exchange (what a ring all-reduce moves ~2 (p-1)/p times per rank)."""
        return 4 * (sum(p.numel() for p in self.big) + self.flat.numel())

    @torch.no_grad()
    def all_reduce_mean(self, world: int, async_op: bool = False, single_rank_too: bool = False):
        """Sum over ranks / world (DDP semantics).  Missing grads count as zeros.  ``single_rank_too`` issues the
        collectives on a one-rank group as well (a no-op in value; tests/test_gpu_rccl.py and bench.py's ``rccl_selftest``
        use it to run the RCCL path on a one-GPU box)."""
        if not dist.is_initialized() or (world <= 1 and not single_rank_too):
            return []
        handles = []
        for p in self.big:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            handles.append(dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, async_op=True))
        off = 0
        for p in self.rest:
            k = p.numel()
            if p.grad is None:
                self.flat[off:off + k].zero_()
            else:
                self.flat[off:off + k].copy_(p.grad.reshape(-1))
            off += k
        if self.flat.numel():
            handles.append(dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=True))
        if async_op:
            return handles
        self.finish(handles, world)
        return []

    @torch.no_grad()
    def finish(self, handles, world: int):
        for h in handles:
            h.wait()
        if not handles:
            return
        for p in self.big:
            p.grad.div_(world)
        off = 0
        for p in self.rest:
            k = p.numel()
            g = (self.flat[off:off + k] / world).view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += k


def spawn_ranks(argv: List[str], n: int, timeout: float | None = None) -> int:
    """Self-launch: start ``n`` copies of ``argv`` (one process per GPU, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set the
    way torchrun sets them; the reference gets its ranks from Lightning's DDP launcher, launch.py:84-97).  The caller
    must not have touched the GPU: children are plain subprocesses, nothing is exec'ed over this process.  Rank 0's
    stdout is passed through; the return value is 0 only if every rank exited 0 (first failing code otherwise, and
    the remaining ranks are terminated)."""
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen(argv, env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    import time
    t0 = time.time()
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in live:          # a dead rank would leave the others waiting in a collective
                    q.terminate()
        if timeout is not None and time.time() - t0 > timeout:
            for q in live:
                q.kill()
            return rc or 124
        time.sleep(0.05)
    sys.stdout.flush()
    return rc


def rccl_selftest(port: int | None = None) -> dict:
    """RCCL on this box, in THIS process (call it in a fresh one: ``python -m rise_sdf_amd.dist --selftest``): a one-rank
    ``nccl`` group on cuda:0 drives the gradient exchange of the data-parallel step (launch.py:84-97, Lightning DDP's all-reduce)
    through GradBuckets on the REAL tensors of the yaml's model -- the 55.4 MiB hash-table gradient and the 18.9 MiB
    environment map in place, every other parameter through the flat buffer.  A one-rank sum is the identity, so the check is
    that the collectives execute on the device (async issue + finish, the way TrainStep uses them) and leave every gradient
    bit-identical.  tests/test_gpu_rccl.py and bench.py's ``secondary.rccl_selftest`` run it."""
    import time
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1")
    if port is not None:
        os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("MASTER_PORT", "29517")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    from . import make
    from .config import tensoir_model_config
    torch.manual_seed(0)
    model = make("split-mixed-occ", tensoir_model_config()).to(dev)      # yaml sizes: L=16 T=2^19, 512^2 cube map
    g = torch.Generator(device=dev).manual_seed(1)
    for p in model.parameters():
        if p.requires_grad:
            p.grad = torch.randn(p.shape, device=dev, generator=g)
    before = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    buckets = GradBuckets(model.parameters())
    res = {"dist_backend": dist.get_backend(), "world": dist.get_world_size(),
           "in_place_tensors": [int(p.numel()) for p in buckets.big], "flat_elements": int(buckets.flat.numel()),
           "bytes_per_step": buckets.bytes_per_step()}
    times = []
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        handles = buckets.all_reduce_mean(1, async_op=True, single_rank_too=True)
        assert len(handles) == len(buckets.big) + 1, len(handles)
        buckets.finish(handles, 1)
        torch.cuda.synchronize()
        times.append(round((time.perf_counter() - t0) * 1e3, 3))
    res["collectives_per_step"] = len(buckets.big) + 1
    res["ms_per_exchange"] = times
    res["bit_identical"] = all(torch.equal(p.grad, before[n]) for n, p in model.named_parameters() if p.grad is not None)
    t = max(model.parameters(), key=lambda p: p.numel()).grad
    ref = t.clone()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    dist.barrier(device_ids=[0])
    torch.cuda.synchronize()
    res["table_sum_identity"] = bool(torch.equal(t, ref))
    dist.destroy_process_group()
    return res


if __name__ == "__main__":
    import json
    import sys
    if "--selftest" in sys.argv:
        print("RESULT " + json.dumps(rccl_selftest()), flush=True)
