"""Data-parallel glue: one process per GPU, torch.distributed (backend "nccl" == RCCL on ROCm) over
xGMI.  The reference has no distributed code at all; the unit that shards is the batch
of (subject, relation) histories: every rank takes its own batch of `batch_size` quadruples with the
reference's per-batch semantics, and the ONE exchange step is an all-reduce of the flat gradient
(80.9 MB fp32 at ICEWS18 sizes) before clip + Adam, i.e. exactly gradient accumulation over
world_size reference batches.

All gradients live in ONE flat fp32 buffer (param.grad are views into it).  xGMI is point-to-point
(7 links x ~153 GB/s per GPU): a ring all-reduce of the 81 MB moves 2*(W-1)/W * 81 MB per GPU ~ 0.9 ms at
W = 8, a fifth of the step, so the exchange is cut into TWO large buckets (few, large collectives suit the
per-link bound) and the first one overlaps the backward pass: the score head's parameters (linear.weight +
bias: 55 of the 81 MB) receive their last gradient contribution when the second entity-head backward has run
-- the FIRST thing the backward pass does -- so their all-reduce is launched on a side stream at that point
(ops.register_grad_done_hook) and runs under the GRU / RGCN backward (~1.5 ms); the rest follows in step().
No measured multi-GPU curve exists yet (one-GPU boxes only): the logic is covered by world-size-2 gloo tests.
"""
import os

import torch
import torch.distributed as dist


def flat_layout(params):
    """Offsets (in floats) of every parameter inside a flat buffer; each starts on a 16-byte boundary so
    that the float4 kernels can address the views directly.  Returns (offsets, total)."""
    offs, off = [], 0
    for p in params:
        offs.append(off)
        off += (p.numel() + 3) & ~3
    return offs, off


def ordered_params(module, first=()):
    """The trainable parameters of `module` in parameter order, those in `first` moved to the front (same relative order)."""
    ps = [p for p in module.parameters() if p.requires_grad]
    ids = {id(p) for p in first}
    return [p for p in ps if id(p) in ids] + [p for p in ps if id(p) not in ids]


class FlatGrads(object):
    """Makes every parameter's .grad a view into one contiguous buffer."""

    def __init__(self, module, first=()):
        """first: parameters to place at the START of the buffer (HipAdam puts the score head there, so that everything
        else is ONE contiguous tail region: two collectives per step instead of three)."""
        self.params = ordered_params(module, first)
        self.offsets, total = flat_layout(self.params)
        dev = self.params[0].device if self.params else torch.device('cpu')
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        for p, off in zip(self.params, self.offsets):
            p.grad = self.flat[off:off + p.numel()].view_as(p)

    def zero(self):
        self.flat.zero_()

    def check_views(self):
        """autograd accumulates in place into an existing .grad; verify nobody replaced the views."""
        base = self.flat.data_ptr()
        for p, off in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != base + 4 * off:
                return False
        return True

    def allreduce_mean(self, group=None):
        """The data path's only collective: average the flat gradient over the ranks."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.flat.div_(dist.get_world_size(group))

    def clip_(self, max_norm):
        """torch.nn.utils.clip_grad_norm_ on the flat buffer (train.py:140)."""
        norm = torch.linalg.vector_norm(self.flat)
        scale = torch.clamp(max_norm / (norm + 1e-6), max=1.0)
        self.flat.mul_(scale)
        return norm

    def span(self, names, module):
        """(offset, length) of the contiguous flat region that holds the named parameters (they must be adjacent
        in parameter order, as linear.weight / linear.bias are)."""
        by_name = {id(p): n for n, p in module.named_parameters()}
        idx = [i for i, p in enumerate(self.params) if by_name.get(id(p)) in names]
        if not idx or idx != list(range(idx[0], idx[-1] + 1)):
            raise ValueError('parameters %r are not adjacent in the flat layout' % (names,))
        lo = self.offsets[idx[0]]
        hi = self.offsets[idx[-1]] + ((self.params[idx[-1]].numel() + 3) & ~3)
        return lo, hi - lo


class OverlapReducer(object):
    """Two-bucket gradient all-reduce with the first bucket overlapped with the backward pass.

    early = the flat region of the parameters whose gradient is complete EARLY in the backward pass (RE-Net: the
    entity score head, model.py:38, whose backward runs first).  A step DECLARES how many in-place accumulations
    complete that bucket (`begin_step(head_passes)`: 2 for the subject + object passes, 1 for the merged pass);
    on_grad_done(param) is called by the autograd Functions right after they accumulated into param.grad
    (ops.grad_done_hooks); when the declared count is reached the bucket's all-reduce is issued asynchronously on a
    side stream (device tensors) / as an async gloo op (CPU tensors) and proceeds while the rest of the backward
    pass runs.  The tail regions (everything else: GRU / RGCN / embeddings, 26 MB at ICEWS18 sizes) are issued
    asynchronously on the same side stream in finish(), i.e. right behind the last backward kernel, so that all
    regions are in flight together; finish() then waits for all of them.  (The tail's gradients are only complete
    when the LAST backward kernel has run -- ent_embeds receives contributions from the very last scatter-add --
    so there is nothing left to overlap it with but the early bucket's own transfer.)

    Safety: a backward pass outside a declared step (smoke / eval with grad, an exception between
    backward and step, more accumulations than declared) never launches early -- the counter only runs between
    begin_step() and finish(), hook calls after the launch raise, and finish() falls back to the synchronous
    exchange whenever the early launch did not happen."""

    def __init__(self, flat_grads, early_span, early_params, early_uses=2, group=None):
        self.fg, self.group = flat_grads, group
        self.lo, self.n = early_span
        self.early_ids = {id(p) for p in early_params}
        self.uses_per_pass = len(self.early_ids)
        self.early_uses = early_uses * self.uses_per_pass
        self.count, self.work, self.armed = 0, None, False
        self.average = True                       # False: SUM (ranks hold disjoint shares of one batch, bench --scaling exact)
        f = flat_grads.flat
        self.early = f[self.lo:self.lo + self.n]
        self.rest = [f[:self.lo], f[self.lo + self.n:]]
        self.stream = torch.cuda.Stream() if f.is_cuda else None

    def begin_step(self, head_passes=2, average=True):
        """Arms the early launch for ONE step: `head_passes` backward passes of the early bucket's parameters will
        run before finish().  Clears any state a previous, unfinished step left behind."""
        if self.work is not None:                 # an abandoned step's collective: complete it before re-arming
            self._wait_early()
        self.early_uses = int(head_passes) * self.uses_per_pass
        self.average = bool(average)
        self.count, self.work, self.armed = 0, None, True

    def set_uses(self, n):
        """Deprecated spelling of begin_step(head_passes=n) (kept for callers of round 2)."""
        self.begin_step(n, self.average)

    def active(self):
        # RENET_FORCE_REDUCER=1 runs the collectives even in a one-rank group (a no-op exchange): lets a one-GPU box
        # exercise the side-stream / RCCL call sequence
        if not (dist.is_available() and dist.is_initialized()):
            return False
        return dist.get_world_size(self.group) > 1 or os.environ.get('RENET_FORCE_REDUCER') == '1'

    def _async(self, t):
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream())          # the accumulating kernels are queued
            with torch.cuda.stream(self.stream):
                return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _wait_early(self):
        self.work.wait()                          # device: makes the current stream wait for the side stream
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        self.work = None

    def on_grad_done(self, param):
        if id(param) not in self.early_ids or not self.armed or not self.active():
            return
        if self.work is not None:
            # the bucket is being reduced on the side stream: one more in-place accumulation into it would race
            raise RuntimeError('OverlapReducer: a backward pass accumulated into the score head\'s gradient after its '
                               'all-reduce was launched (begin_step(head_passes=%d) declared too few passes)'
                               % (self.early_uses // max(self.uses_per_pass, 1)))
        self.count += 1
        if self.count == self.early_uses:
            self.work = self._async(self.early)

    def finish(self, fold_scale=False):
        """Call after backward(): completes the exchange; the flat buffer then holds the rank-combined gradient:
        the SUM when `average` is False, else the mean -- divided in place here, or (fold_scale=True, HipAdam) left
        as the sum with `pending_scale` = 1 / world for the optimizer kernel to apply (no extra pass over 81 MB)."""
        armed, self.armed = self.armed, False
        self.pending_scale = 1.0
        if not self.active():
            self.count, self.work = 0, None
            return
        world = dist.get_world_size(self.group)
        if self.work is not None and self.count != self.early_uses:      # cannot happen (the hook raises); belt and braces
            raise RuntimeError('OverlapReducer: early bucket launched after %d of %d accumulations'
                               % (self.count, self.early_uses))
        # tail bucket(s): asynchronously as well, so that the two regions' transfers are both in flight
        tails = [self._async(t) for t in self.rest if t.numel()]
        if self.work is None:                     # never launched early (undeclared step, empty batches, fewer passes)
            tails.append(self._async(self.early))
        else:
            self._wait_early()
        for w in tails:
            w.wait()
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        if self.average:
            if fold_scale:
                self.pending_scale = 1.0 / world
            else:
                self.fg.flat.div_(world)
        self.count, self.work = 0, None
        del armed


def shard_indices(perm, step, rank, world, batch_size):
    """Rank `rank`'s quadruple indices for global step `step`: consecutive batch_size slices of the
    shuffled order are dealt round-robin to the ranks (weak scaling: per-rank batch is fixed)."""
    k = step * world + rank
    n = len(perm)
    start = (k * batch_size) % max(n - batch_size + 1, 1)
    return perm[start:start + batch_size]


class ScalingMode(object):
    """How ONE data-parallel step divides its quadruples over the ranks and combines their gradients -- the bookkeeping
    bench.py's three modes (and a training loop) share, kept here so that it is tested without a GPU
    (tests/test_parallel_cpu.py):

      weak   : every rank takes its own `batch` quadruples (global batch = batch * world), each rank builds its own batch
               graph, gradients are AVERAGED (SURVEY 8e option (ii): the reference's semantics at batch size `batch`)
      strong : `batch` quadruples per step in total, batch // world per rank (each rank its own, smaller, reference batch;
               global batch = (batch // world) * world), gradients averaged
      exact  : ONE reference batch of `batch` quadruples per step; every rank builds ITS graph (same on every rank) and keeps
               1 / world of the sequences (graph.shard_sequences), gradients and losses are SUMMED -- the N-rank step
               equals the 1-rank step on that batch (SURVEY 8e option (i)); reported as scaling "strong"."""

    MODES = ('weak', 'strong', 'exact')

    def __init__(self, scaling, batch, world, passes='merged'):
        if scaling not in self.MODES:
            raise ValueError('scaling must be one of %s' % (self.MODES,))
        self.scaling, self.world = scaling, int(world)
        self.exact = scaling == 'exact'
        self.rank_batch = int(batch) if scaling in ('weak', 'exact') else max(1, int(batch) // self.world)
        self.global_batch = int(batch) if self.exact else self.rank_batch * self.world
        self.average = not self.exact                 # exact: per-rank losses are partial sums of the batch mean
        self.passes = 'merged' if self.exact else passes
        self.reported_scaling = 'strong' if scaling in ('strong', 'exact') else 'weak'

    def indices(self, perm, step, rank):
        """The quadruple indices rank `rank` builds its batch from at global step `step`."""
        if self.exact:
            return shard_indices(perm, step, 0, 1, self.rank_batch)          # the same reference batch on every rank
        return shard_indices(perm, step, rank, self.world, self.rank_batch)

    def shard(self, rank):
        """`shard=` argument of RENet.prepare_both: which sequences of the shared batch this rank keeps (exact mode)."""
        return (rank, self.world) if (self.exact and self.world > 1) else None


class FlatParams(object):
    """Moves every parameter of `module` into ONE contiguous buffer (param.data become views), in the same
    order as FlatGrads, so that the optimizer is a single fused kernel over flat buffers."""

    def __init__(self, module, first=()):
        self.params = ordered_params(module, first)
        self.offsets, total = flat_layout(self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.numel = total
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                n = p.numel()
                self.flat[off:off + n].copy_(p.data.reshape(-1))
                p.data = self.flat[off:off + n].view_as(p)


class HipAdam(object):
    """torch.optim.Adam(lr, weight_decay) + clip_grad_norm_(max_norm) + zero_grad as ONE fused HIP step on
    the flat buffers (renet_adam_step).  Matches train.py:61,140-142.

    Data-parallel use: wrap every step in `with opt.step_scope(head_passes=...)` (declares how many backward passes
    of the score head the step runs, so that its 55 MB gradient bucket can be all-reduced under the rest of the
    backward pass); a step outside a scope still works -- the whole exchange then happens inside step()."""

    def __init__(self, module, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_norm=0.0):
        import renet_hip as K
        self.K = K
        self._module = module          # (its `gemm_mode` attribute is read at every step)
        # the score head's parameters FIRST in both flat buffers: the early bucket is then [0, n) and everything else one
        # contiguous tail -- two collectives per step instead of three (round 5; profiles/r05_i_one_rank_rccl_trace.md)
        head = [p for n, p in module.named_parameters() if n in ('linear.weight', 'linear.bias')]
        self.params = FlatParams(module, first=head)
        self.grads = FlatGrads(module, first=head)     # same layout (flat_layout) as the parameters
        self.m = torch.zeros_like(self.params.flat)
        self.v = torch.zeros_like(self.params.flat)
        self.lr, self.betas, self.eps, self.wd, self.max_norm = lr, betas, eps, weight_decay, max_norm
        self.t = 0
        self.norm = torch.zeros(1, device=self.m.device, dtype=torch.float32)
        self.reducer = None
        self._hook = None
        # bf16-storage mode: the GEMM wrappers may cache bf16 copies of these weights between steps
        self._weights = [p for p in self.params.params if p.dim() == 2]
        K.register_weights(self._weights)
        names = [n for n, _ in module.named_parameters()]
        if 'linear.weight' in names and 'linear.bias' in names:       # RENet: overlap the score head's bucket
            early = [p for n, p in module.named_parameters() if n in ('linear.weight', 'linear.bias')]
            self.reducer = OverlapReducer(self.grads, self.grads.span(('linear.weight', 'linear.bias'), module), early)
            # registry keyed by parameter identity (a second optimizer must not steal a global hook);
            # close() / garbage collection of this optimizer unregisters
            import ops
            self._hook = ops.register_grad_done_hook(early, self.reducer.on_grad_done)

    def close(self):
        """Unregisters the gradient hooks (the reducer, the flat buffers and the parameters are released with it)."""
        if self._hook is not None:
            import ops
            ops.unregister_grad_done_hook(self._hook)
            self._hook = None
        if getattr(self, '_weights', None):
            self.K.unregister_weights(self._weights)
            self._weights = []

    def __del__(self):
        try:
            self.close()
        except Exception:       # interpreter shutdown
            pass

    def step_scope(self, head_passes=2, average=True):
        """Context manager around ONE training step (forward, backward, step()).  Inside it the parameter-gradient
        kernels that feed nothing but the optimizer may still be running on a side stream when backward() returns
        (ops.DEFER_WEIGHT_GRADS): step() waits for them; code that reads `.grad` BEFORE step() -- gradient logging, a
        custom clip -- calls sync_grads() first."""
        return _StepScope(self, head_passes, average)

    @staticmethod
    def sync_grads():
        """Orders the current stream behind every deferred gradient kernel (no-op when nothing is pending)."""
        import ops
        ops.join_deferred()

    def step(self):
        """all-reduce (if distributed; the score head's bucket was started during backward) -> clip -> Adam ->
        zero_grad.  The 1/world of the gradient average is folded into the optimizer kernel (no pass over 81 MB)."""
        scale = 1.0
        import ops
        ops.join_deferred()                  # deferred side-stream gradient work (ops.DEFER_WEIGHT_GRADS) ends here
        if self.reducer is not None:
            self.reducer.finish(fold_scale=True)
            scale = self.reducer.pending_scale
        elif dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.grads.flat, op=dist.ReduceOp.SUM)
            scale = 1.0 / dist.get_world_size()
        self.t += 1
        self.K.adam_step(self.params.flat, self.grads.flat, self.m, self.v, self.lr, self.betas[0], self.betas[1],
                         self.eps, self.wd, self.max_norm, self.t, True, self.norm, grad_scale=scale)
        self.K.weights_changed()
        # f16x3 mode: the magnitude bounds of all registered weights, measured HERE on the step's stream (one launch), so
        # that no GEMM of the next step -- on whichever stream ops._Side puts it -- is the one that triggers the pass
        with self.K.gemm_mode(getattr(self._module, 'gemm_mode', None)):
            self.K.prefetch_weight_bounds(self.params.flat.device)


class _StepScope(object):
    def __init__(self, opt, head_passes, average):
        self.opt, self.head_passes, self.average = opt, head_passes, average

    def __enter__(self):
        if self.opt.reducer is not None:
            self.opt.reducer.begin_step(self.head_passes, self.average)
        # inside a declared step the optimizer's step() is the one consumer of the parameter gradients: side-stream
        # work that only feeds them may stay un-joined until then (ops.DEFER_WEIGHT_GRADS; RENET_DEFER_GRADS=0 turns it off)
        import ops
        self._defer_old = ops.DEFER_WEIGHT_GRADS
        ops.DEFER_WEIGHT_GRADS = os.environ.get('RENET_DEFER_GRADS', '1') != '0'
        return self.opt

    def __exit__(self, exc_type, exc, tb):
        import ops
        ops.DEFER_WEIGHT_GRADS = self._defer_old
        ops.join_deferred()                  # (a step abandoned before step(): nothing stays pending)
        r = self.opt.reducer
        if r is not None and r.armed:
            # the step was abandoned before step() (exception, early exit): complete a launched collective so that
            # no rank is left with a pending RCCL op, and disarm
            r.armed = False
            if r.work is not None:
                r._wait_early()
            r.count = 0
        return False
