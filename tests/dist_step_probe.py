#!/usr/bin/env python3
"""Rank body of tests/test_gpu_dist_step.py (launched by rise_sdf_amd.dist.spawn_ranks, two ranks on one GPU with
RSDF_DIST_SHARE_GPU=1): the config[3]-shaped training step -- occupancy update on every rank, each rank's own ray batch
through the occupancy-pruned sampler, backward, the DDP-style gradient mean (GradBuckets), Adam -- and the checks that
make it a correct data-parallel step (launch.py:84-97):

  1. the averaged gradient equals the mean of the two ranks' gradients recomputed in ONE process on the same batches;
  2. after 3 steps the parameters, the optimizer's view of them and ``occupancy_grid.binaries`` / ``occs`` are bit-identical
     on both ranks.
Rank 0 prints ``RESULT {json}``."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(dev, rank, world, stage):
    import rise_sdf_amd as R
    from rise_sdf_amd.config import TENSOIR_LAMBDAS, TENSOIR_REG_LAMBDAS, tensoir_model_config, tensoir_optimizer
    from rise_sdf_amd.dist import GradBuckets
    from rise_sdf_amd.step import TrainStep
    from rise_sdf_amd.synthetic import make_dataset
    torch.manual_seed(0)                                  # identical initial replicas
    cfg = tensoir_model_config(hidden=64, log2_T=15, n_levels=8, split_sum_kick_in_step=0 if stage else 1 << 60,
                               train_num_rays=512, max_train_num_rays=1024)
    cfg["light"]["envlight_config"]["base_res"] = 64
    model = R.make("split-mixed-occ", cfg).to(dev)
    model.train()
    ds = make_dataset(n_views=4, W=96, H=96, seed=0, device=dev)
    opt = tensoir_optimizer(model)
    buckets = GradBuckets(model.parameters())
    ts = TrainStep(model, ds, opt, TENSOIR_LAMBDAS, train_num_rays=512, max_train_num_rays=1024, num_samples_per_ray=256,
                   seed=0, rank=rank, world=world, grad_buckets=buckets, reg_lambdas=TENSOIR_REG_LAMBDAS)
    return model, ts, buckets


def grads_of(model):
    return {n: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for n, p in model.named_parameters()}


def main():
    from rise_sdf_amd import dist as rdist
    rank, local, world = rdist.init_from_env()
    assert world == 2
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    stage = int(os.environ.get("PROBE_STAGE", "1"))
    model, ts, buckets = build(dev, rank, world, stage)
    gs0 = 20000                                            # all levels active, eps = one finest cell
    res = {}

    # ---- 1. one backward: all-reduced gradient vs the single-process mean over both ranks' batches ------------------
    model.update_step(0, gs0)                              # occupancy update (step % 16 == 0) on every rank
    batch = ts.sample_batch()
    torch.manual_seed(1000 + rank)                         # stratified jitter / curvature directions of this forward
    ts.forward_backward(batch)
    own = grads_of(model)
    buckets.all_reduce_mean(world)
    avg = grads_of(model)
    if rank == 0:
        from rise_sdf_amd.step import TrainStep
        per_rank = []
        for r in range(world):
            tr = TrainStep(model, ts.ds, ts.opt, ts.lambdas, train_num_rays=512, max_train_num_rays=1024,
                           num_samples_per_ray=256, seed=0, rank=r, world=1, reg_lambdas=ts.reg_lambdas)
            b = tr.sample_batch()                          # a fresh generator replays rank r's first draw
            torch.manual_seed(1000 + r)
            tr.forward_backward(b)
            per_rank.append(grads_of(model))
        worst, worst_own = 0.0, 0.0
        for n in avg:
            ref = (per_rank[0][n] + per_rank[1][n]) / 2
            scale = float(ref.abs().max()) + 1e-20
            worst = max(worst, float((avg[n] - ref).abs().max()) / scale)
            worst_own = max(worst_own, float((own[n] - per_rank[0][n]).abs().max()) / (float(per_rank[0][n].abs().max()) + 1e-20))
        res["allreduce_vs_single_process_rel"] = worst
        res["replay_of_own_batch_rel"] = worst_own
        res["ranks_drew_different_rays"] = bool((per_rank[0]["variance.variance"] != per_rank[1]["variance.variance"]).any())
    dist.barrier()

    # ---- 2. three optimizer steps: replicas stay bit-identical ---------------------------------------------------------
    samples = []
    for k in range(3):
        r = ts.step(gs0 + 16 * (k + 1))                    # every step a grid-update step
        samples.append(r["num_samples"])
    state = [p.detach().reshape(-1).view(torch.int32) for p in model.parameters()]
    state += [model.occupancy_grid.binaries.reshape(-1).to(torch.int32), model.occupancy_grid.occs.view(torch.int32)]
    flat = torch.cat(state)
    other = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(other, flat)
    if rank == 0:
        res["replicas_bit_identical"] = bool(torch.equal(other[0], other[1]))
        if not res["replicas_bit_identical"]:              # name the tensors that differ
            names = [n for n, _ in model.named_parameters()] + ["occupancy_grid.binaries", "occupancy_grid.occs"]
            off, bad = 0, []
            for n, t in zip(names, state):
                k = t.numel()
                d = int((other[0][off:off + k] != other[1][off:off + k]).sum())
                if d:
                    bad.append((n, d, k))
                off += k
            res["differing"] = bad[:12]
        res["n_state_words"] = int(flat.numel())
        res["occupied_cells"] = int(model.occupancy_grid.binaries.sum())
        res["samples_rank0"] = samples
        res["stage"] = int(model.stage)
        print("RESULT " + json.dumps(res), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
