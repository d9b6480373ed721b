#!/usr/bin/env python3
"""Body of tests/test_gpu_rccl.py: a fresh process that initialises ``backend='nccl'`` (= RCCL on ROCm) as a one-rank group
on cuda:0 and drives the gradient exchange of the data-parallel step (launch.py:84-97, Lightning DDP's all-reduce) through
rise_sdf_amd.dist.GradBuckets on the REAL tensors: the 55.4 MiB hash-table gradient and the 18.9 MiB environment map in
place, every other parameter through the flat buffer.  A one-rank sum is the identity, so the check is that the collectives
execute on the device (async issue + finish, the way TrainStep uses them) and leave every gradient bit-identical.
Prints ``RESULT {json}``."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")       # (the test passes a free port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    import rise_sdf_amd as R
    from rise_sdf_amd.config import tensoir_model_config
    from rise_sdf_amd.dist import GradBuckets
    torch.manual_seed(0)
    model = R.make("split-mixed-occ", tensoir_model_config()).to(dev)      # yaml sizes: L=16 T=2^19, 512^2 cube map
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
    for it in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        handles = buckets.all_reduce_mean(1, async_op=True, single_rank_too=True)
        assert len(handles) == len(buckets.big) + 1, len(handles)
        buckets.finish(handles, 1)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    res["collectives_per_step"] = len(buckets.big) + 1
    res["ms_per_exchange"] = times
    res["bit_identical"] = all(torch.equal(p.grad, before[n]) for n, p in model.named_parameters() if p.grad is not None)
    # a plain sum over the group on the table gradient too (what all_reduce_mean issues, checked by value)
    t = max(model.parameters(), key=lambda p: p.numel()).grad
    ref = t.clone()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    dist.barrier(device_ids=[0])
    torch.cuda.synchronize()
    res["table_sum_identity"] = bool(torch.equal(t, ref))
    print("RESULT " + json.dumps(res), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
