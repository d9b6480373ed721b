"""The training step around the hot path, as ``systems/split_occ.py`` drives it (the Lightning system
itself is out of scope, its per-step arithmetic is what config[3] / config[4] of BASELINE.json exercise):

  on_train_batch_start   systems/base.py:81-84          update_module_step(model, epoch, global_step)
  preprocess_data        systems/split_occ.py:58-131    random (image, y, x) -> rays, rgb, fg_mask; background colour
  training_step          systems/split_occ.py:150-237   build_mips, forward, dynamic_ray_sampling, loss terms
  backward / optimizer   Lightning + DDP                gradient mean all-reduce over the ranks (launch.py:84-97)

``TrainStep.step`` issues all of it on the current stream: ray generation + pixel gather in one kernel
(``rsdf_gen_rays``, N2), the loss tail in two reduction + two elementwise kernels (N1), gradients averaged through
``rise_sdf_amd.dist.GradBuckets``.  Host reads per step: the sample counts the model's sampler already brings back
(``num_samples_host``); ``dynamic_ray_sampling`` reuses that value instead of ``.item()``-ing the output tensor again
(systems/split_occ.py:160).
"""
from __future__ import annotations

import torch

from . import ops
from .loss import loss_tail


class TrainStep:
    def __init__(self, model, dataset, optimizer, lambdas, *, train_num_rays=256, max_train_num_rays=4096,
                 num_samples_per_ray=1024, dynamic_ray_sampling=True, background_color="random", apply_mask=True,
                 sparsity_scale=1.0, seed=0, rank=0, world=1, grad_buckets=None, reg_lambdas=None, sync_free=True,
                 prefilter_on_side_stream=True):
        self.model, self.ds, self.opt, self.lambdas = model, dataset, optimizer, dict(lambdas)
        self.train_num_rays = int(train_num_rays)
        self.max_train_num_rays = int(max_train_num_rays)
        # systems/split_occ.py:51: the sample budget the ray count is steered to
        self.train_num_samples = int(train_num_rays) * int(num_samples_per_ray)
        self.dynamic_ray_sampling = bool(dynamic_ray_sampling)
        self.background_color, self.apply_mask = background_color, bool(apply_mask)
        self.sparsity_scale = float(sparsity_scale)
        self.world, self.buckets = int(world), grad_buckets
        self.reg_lambdas = dict(reg_lambdas or {})
        dev = dataset["all_images"].device
        # the reference seeds every rank identically (launch.py:63-65): seed + rank here
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(int(seed) + int(rank))
        self.dev = dev
        # The occupancy grid is a replicated buffer.  The reference keeps the replicas equal through DDP's per-forward
        # buffer broadcast from rank 0; here every rank runs the same update on the same (all-reduced)
        # parameters with the same random cells and jitter, from a generator of its own that every rank seeds alike and
        # that nothing else draws from (SURVEY 8e option (a): no traffic).
        grid = getattr(model, "occupancy_grid", None)
        self.sync_free = bool(sync_free)
        import os
        if os.environ.get("RSDF_PREFILTER_STREAM", "") == "main":      # A/B knob: the prefilter on the step's own stream
            prefilter_on_side_stream = False
        self.prefilter_stream = (torch.cuda.Stream(device=dev) if (prefilter_on_side_stream and sync_free
                                                                   and dev.type == "cuda") else None)
        if sync_free:
            # Host reads of the reference's step: marcher total + compaction count per sampling call (x2 with
            # secondary rays), torch.nonzero for the secondary rays, num_samples.item() for dynamic_ray_sampling = 6, plus
            # two host-to-device copies this mirror had added.  Here: ONE read per sampling call (capacity mode,
            # nerfacc/__init__.py) = 2 per step with secondary rays, 1 without; identical samples and values.
            if grid is not None and hasattr(grid, "capacity_mode"):
                grid.capacity_mode = True
            if hasattr(model, "masked_secondary"):
                model.masked_secondary = True
        if dev.type == "cuda":
            from . import _lib
            self._st_seen = _lib.status(dev).clone()                 # the status words as of the last optimizer step
        if grid is not None and getattr(grid, "rng", 0) is None:     # (a generator already in place is kept)
            grid.rng = torch.Generator(device=dev)
            grid.rng.manual_seed(int(seed) + 7919)

    def sample_batch(self, n=None):
        """systems/split_occ.py:58-131 (training branch, batch_image_sampling)."""
        n = self.train_num_rays if n is None else int(n)
        ds, dev = self.ds, self.dev
        V = ds["all_images"].shape[0]
        index = torch.randint(0, V, (n,), device=dev, generator=self.gen)
        x = torch.randint(0, ds["w"], (n,), device=dev, generator=self.gen)
        y = torch.randint(0, ds["h"], (n,), device=dev, generator=self.gen)
        if self.background_color == "random":
            bg = torch.rand(3, device=dev, generator=self.gen)
        else:
            bg = torch.full((3,), 1.0 if self.background_color == "white" else 0.0, device=dev)
        rays, rgb, fg = ops.gen_rays(index, y, x, ds["directions"], ds["all_c2w"], ds["all_images"],
                                     ds["all_fg_masks"], bg, apply_mask=self.apply_mask)
        return {"rays": rays, "rgb": rgb, "fg_mask": fg, "background_color": bg}

    def forward_backward(self, batch):
        """build_mips, forward, loss tail + regularisations, backward (systems/split_occ.py:150-237 and Lightning's
        backward): gradients are left in ``.grad``, not yet averaged over ranks.  -> (loss, terms, out)."""
        model = self.model
        model.background_color = batch["background_color"]
        if getattr(model, "emitter", None) is not None and getattr(model, "stage", 0):
            # systems/split_occ.py:151-152.  The prefilter does not depend on the samples and nothing before the render
            # stage reads it, so it is issued right AFTER the model's sampling call (whose host read it would otherwise
            # only delay; split_mixed_occ.forward_): same values, the host gets the prefilter's kernel time as a head start
            if self.sync_free and hasattr(model, "after_sampling"):
                if self.prefilter_stream is not None and hasattr(model.emitter, "build_mips_on"):
                    # ... and on a side stream: vector-ALU work beside the networks' matrix / memory work, forward and
                    # (autograd runs a node's backward on its forward's stream) backward
                    side, em = self.prefilter_stream, model.emitter
                    model.after_sampling = lambda: em.build_mips_on(side)
                else:
                    model.after_sampling = model.emitter.build_mips
            else:
                model.emitter.build_mips()
        out = model(batch["rays"])
        if getattr(model, "after_sampling", None) is not None:   # a forward that never reached its sampling call
            model.after_sampling = None
            model.emitter.build_mips()
        loss, terms = loss_tail(out, batch, self.lambdas, sparsity_scale=self.sparsity_scale)
        for name, value in model.regularizations(out).items():    # systems/split_occ.py:217-221
            lam = self.reg_lambdas.get("lambda_" + name, 0.0)
            if lam:
                loss = loss + lam * value
        self.opt.zero_grad(set_to_none=True)
        loss.backward()
        return loss, terms, out

    def _guarded_optimizer_step(self):
        """the render pass runs AFTER the step's last host read, so a forward range violation of the two-part fp16
        kernels (inf / nan outputs -> nan loss -> nan gradients) would reach Adam before any poll sees the status words, and
        one step destroys every parameter and moment.  The optimizer step is therefore made conditional ON THE DEVICE: the
        status word's movement since the previous step becomes ``found_inf`` of torch's fused Adam (the GradScaler hook: the
        kernel leaves parameters, moments and step counts untouched when it is set) -- no host read.  The next host read
        (the next step's sampler) then switches the offending kernel family to the range-free kernels (_lib.poll_status), so
        the run continues where the reference's fp32 networks would have.  Optimizers without that hook pay one 32-byte read."""
        from . import _lib
        if self.dev.type != "cuda":
            self.opt.step()
            return
        st = _lib.status(self.dev)
        if all(getattr(g, "get", lambda *_: False)("fused", False) for g in self.opt.param_groups):
            found = (st[_lib.ST_X2_FWD_NONFINITE] != self._st_seen[_lib.ST_X2_FWD_NONFINITE]).to(torch.float32).reshape(())
            self._st_seen.copy_(st)
            if self.world > 1 and torch.distributed.is_initialized():      # every replica skips, or none
                torch.distributed.all_reduce(found, op=torch.distributed.ReduceOp.MAX)
            self.opt.grad_scale, self.opt.found_inf = None, found
            try:
                self.opt.step()
            finally:
                self.opt.found_inf = None
            self.last_found_inf = found                # (a device tensor: reading it is the caller's host read)
        else:
            r = _lib.poll_status(self.dev)
            if r["x2_fwd_nonfinite"]:
                self.opt.zero_grad(set_to_none=True)   # that forward was not the reference's: skip, the next one is range-free
                self.skipped_steps = getattr(self, "skipped_steps", 0) + 1
                return
            self.opt.step()

    def step(self, global_step, batch=None, epoch=0):
        self.model.update_step(epoch, global_step)                # systems/base.py:81-84
        if batch is None:
            batch = self.sample_batch()
        loss, terms, out = self.forward_backward(batch)
        handles = []
        if self.world > 1 and self.buckets is not None:           # DDP: mean of the ranks' gradients (launch.py:84-97):
            handles = self.buckets.all_reduce_mean(self.world, async_op=True)   # issued here, finished just before Adam
        n_rays = batch["rays"].shape[0]
        ns = out.get("num_samples_host", None)
        if ns is None:                                            # a model that did not read its count back
            ns = int(out["num_samples_full"].sum().item())
        if self.dynamic_ray_sampling and ns > 0:                  # systems/split_occ.py:159-161
            want = int(self.train_num_rays * (self.train_num_samples / ns))
            self.train_num_rays = min(int(self.train_num_rays * 0.9 + want * 0.1), self.max_train_num_rays)
        if handles:
            self.buckets.finish(handles, self.world)
        self._guarded_optimizer_step()
        res = {"loss": loss.detach(), "terms": terms, "num_samples": ns, "num_rays": n_rays, "out": out}
        # a read-free (blind) secondary sampling pass that outgrew its buffers ran truncated (its tail rays unoccluded): the
        # caller can skip or redo the step instead of relying on the sampler's RuntimeWarning (shown once per location)
        grid = getattr(self.model, "occupancy_grid", None)
        st = getattr(grid, "stats", None)
        if st is not None:
            n_over = st.get("blind_overflows", 0)
            if n_over != getattr(self, "_seen_blind_overflows", 0):
                res["blind_overflow"] = st.get("last_blind_overflow")
                self._seen_blind_overflows = n_over
        return res


def build_synthetic_training(dev, *, stage=1, hidden=128, views=8, res=200, seed=0, rank=0, world=1, indirect=True,
                             curvature=True, grad_buckets=True, model_overrides=None, tex_precision="fp32",
                             sdf_precision="fp32", fused_adam=True):
    """The full split-mixed-occ model at the yaml's sizes (rise_sdf_amd.config.tensoir_model_config) with the yaml's
    optimizer and loss weights, on the synthetic analytic scene (rise_sdf_amd.synthetic): what bench.py's config[3]
    workload, tools/bench_step.py and the multi-rank tests drive.  -> (model, TrainStep)."""
    from . import make
    from .config import TENSOIR_LAMBDAS, TENSOIR_REG_LAMBDAS, tensoir_model_config, tensoir_optimizer
    from .dist import GradBuckets
    from .synthetic import make_dataset
    torch.manual_seed(seed)                                   # identical initial replicas on every rank
    cfg = tensoir_model_config(hidden=hidden, indirect_pred=indirect, curvature=curvature,
                               tex_precision=tex_precision, sdf_precision=sdf_precision,
                               split_sum_kick_in_step=0 if stage else 1 << 60, **(model_overrides or {}))
    model = make("split-mixed-occ", cfg).to(dev)
    model.train()
    ds = make_dataset(n_views=views, W=res, H=res, seed=seed, device=dev)
    opt = tensoir_optimizer(model, fused=fused_adam)   # (fused multi-tensor Adam: ~30 launches less per step)
    lam = dict(TENSOIR_LAMBDAS)
    if not curvature:
        lam["lambda_curvature"] = 0.0
    buckets = GradBuckets(model.parameters()) if (grad_buckets and world > 1) else None
    ts = TrainStep(model, ds, opt, lam, train_num_rays=cfg.train_num_rays, max_train_num_rays=cfg.max_train_num_rays,
                   num_samples_per_ray=cfg.num_samples_per_ray, dynamic_ray_sampling=True, seed=seed, rank=rank,
                   world=world, grad_buckets=buckets, reg_lambdas=TENSOIR_REG_LAMBDAS)
    return model, ts
