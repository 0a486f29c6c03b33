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
(ops.register_grad_done_hook) and runs under the GRU / RGCN backward (~1.5 ms).  Round 6: a THIRD bucket, the two
encoders' parameters (4.5 MB), leaves when the GRU parameter-gradient GEMMs are queued (under the sequence-assembly /
RGCN backward); the rest follows in step(); each region's sum of squares is accumulated right behind its all-reduce on
the reducer's stream, so that the clip of train.py:140 only waits for a scalar combine after the last bucket.
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
    """The trainable parameters of `module` in parameter order, those in `first` moved to the front IN THE ORDER GIVEN."""
    ps = [p for p in module.parameters() if p.requires_grad]
    own = {id(p) for p in ps}
    front, seen = [], set()
    for p in first:
        if id(p) in own and id(p) not in seen:
            front.append(p)
            seen.add(id(p))
    return front + [p for p in ps if id(p) not in seen]


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


class _Bucket(object):
    """One region of the flat gradient whose all-reduce is launched as soon as its last accumulation has been queued."""

    def __init__(self, name, span, params, flat):
        self.name = name
        self.lo, self.n = span
        self.ids = {id(p) for p in params}
        self.uses_per_pass = len(self.ids)
        self.expected = 0
        self.count, self.work = 0, None
        self.view = flat[self.lo:self.lo + self.n]


class OverlapReducer(object):
    """Bucketed gradient all-reduce with the early buckets overlapped with the backward pass.

    early = the flat region of the parameters whose gradient is complete EARLY in the backward pass (RE-Net: the
    entity score head, model.py:38, whose backward runs first); mid (optional, round 6) = the two encoders' parameters
    (model.py:28-29), complete when the GRU parameter-gradient GEMMs have run, i.e. before the sequence-assembly / RGCN
    backward and the embedding scatter-adds.  A step DECLARES how many in-place accumulations complete a bucket
    (`begin_step(head_passes)`: 2 for the subject + object passes, 1 for the merged pass); on_grad_done(param) is called
    by the autograd Functions right after they accumulated into param.grad (ops.grad_done_hooks) -- on the stream that
    holds the accumulating kernels -- and when a bucket's declared count is reached its all-reduce is issued
    asynchronously on a side stream (device tensors) / as an async gloo op (CPU tensors) and proceeds while the rest of
    the backward pass runs.  The tail regions (everything else: RGCN / embeddings / relation head) are issued in finish(),
    i.e. right behind the last backward kernel; finish() then waits for all of them.  (The tail's gradients are only
    complete when the LAST backward kernel has run -- ent_embeds receives contributions from the very last scatter-add.)

    Per-region sums of squares (round 6): `sumsq_fn(k, region)` -- when given -- is called for region k of `regions()`
    right behind that region's all-reduce, in stream order on the reducer's stream, so that by the time the last bucket
    arrives only a scalar combine is left of clip_grad_norm_ (train.py:140); `partials_ready` tells the optimizer whether
    every region of this step went through it.  CPU tensors: `bucket_sumsq[k]` holds the region's sum of squares.

    Safety: a backward pass outside a declared step (smoke / eval with grad, an exception between
    backward and step, more accumulations than declared) never launches early -- the counters only run between
    begin_step() and finish(), hook calls after a bucket's launch raise, and finish() falls back to the synchronous
    exchange for whatever was not launched early."""

    def __init__(self, flat_grads, early_span, early_params, early_uses=2, group=None, mid_span=None, mid_params=(),
                 sumsq_fn=None):
        self.fg, self.group = flat_grads, group
        f = flat_grads.flat
        self.buckets = [_Bucket('early', early_span, early_params, f)]
        if mid_span is not None and mid_span[1] > 0:
            self.buckets.append(_Bucket('mid', mid_span, mid_params, f))
        self.buckets.sort(key=lambda b: b.lo)
        for x, y in zip(self.buckets, self.buckets[1:]):
            if x.lo + x.n > y.lo:
                raise ValueError('OverlapReducer: overlapping buckets')
        # the tail regions: what lies between / around the timed buckets, in flat order
        self.rest, pos = [], 0
        for bk in self.buckets:
            if bk.lo > pos:
                self.rest.append((pos, bk.lo - pos))
            pos = bk.lo + bk.n
        if pos < f.numel():
            self.rest.append((pos, f.numel() - pos))
        self.armed = False
        self.average = True                       # False: SUM (ranks hold disjoint shares of one batch, bench --scaling exact)
        self.stream = torch.cuda.Stream() if f.is_cuda else None
        self.sumsq_fn = sumsq_fn
        self.partials_ready = False
        self.bucket_sumsq = {}
        self._done_regions = set()
        self.begin_uses(early_uses)

    # ---- compatibility with the two-bucket reducer of rounds 2-5 (tests, tools) -------------------------------
    @property
    def early(self):
        return self._bucket('early').view

    @property
    def work(self):
        return self._bucket('early').work

    @work.setter
    def work(self, v):
        self._bucket('early').work = v

    @property
    def count(self):
        return self._bucket('early').count

    @count.setter
    def count(self, v):
        for bk in self.buckets:
            bk.count = v

    @property
    def early_uses(self):
        return self._bucket('early').expected

    def _bucket(self, name):
        for bk in self.buckets:
            if bk.name == name:
                return bk
        raise KeyError(name)

    def regions(self):
        """[(offset, length)] of every region the exchange handles, in flat order (timed buckets and tails)."""
        return sorted([(bk.lo, bk.n) for bk in self.buckets] + list(self.rest))

    def begin_uses(self, passes):
        for bk in self.buckets:
            bk.expected = int(passes) * bk.uses_per_pass

    def begin_step(self, head_passes=2, average=True):
        """Arms the early launches for ONE step: `head_passes` backward passes of the timed buckets' parameters will
        run before finish().  Clears any state a previous, unfinished step left behind."""
        for bk in self.buckets:
            if bk.work is not None:               # an abandoned step's collective: complete it before re-arming
                self._wait(bk)
        self.begin_uses(head_passes)
        self.average = bool(average)
        for bk in self.buckets:
            bk.count, bk.work = 0, None
        self.armed = True
        self.partials_ready = False
        self._done_regions = set()

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

    def _after(self, work, lo, n):
        """Orders the caller behind `work` and runs the region's sum of squares right behind it."""
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                work.wait()                       # (device: the reducer's stream waits for the collective, the host does not)
                if self.sumsq_fn is not None:
                    self.sumsq_fn(self.regions().index((lo, n)), self.fg.flat[lo:lo + n])
                    self._done_regions.add((lo, n))
            return
        work.wait()
        t = self.fg.flat[lo:lo + n]
        self.bucket_sumsq[self.regions().index((lo, n))] = float((t.double() * t.double()).sum())
        self._done_regions.add((lo, n))

    def _wait(self, bk):
        self._after(bk.work, bk.lo, bk.n)
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        bk.work = None

    def _wait_early(self):                        # (round 2-5 name)
        self._wait(self._bucket('early'))

    def on_grad_done(self, param):
        if not self.armed:
            return
        pid = id(param)
        for bk in self.buckets:
            if pid in bk.ids:
                break
        else:
            return
        if not self.active():
            return
        if bk.work is not None:
            # the bucket is being reduced on the side stream: one more in-place accumulation into it would race
            raise RuntimeError('OverlapReducer: a backward pass accumulated into the %s bucket\'s gradient after its '
                               'all-reduce was launched (begin_step(head_passes=%d) declared too few passes)'
                               % (bk.name, bk.expected // max(bk.uses_per_pass, 1)))
        bk.count += 1
        if bk.count == bk.expected:
            bk.work = self._async(bk.view)

    def finish(self, fold_scale=False):
        """Call after backward(): completes the exchange; the flat buffer then holds the rank-combined gradient:
        the SUM when `average` is False, else the mean -- divided in place here, or (fold_scale=True, HipAdam) left
        as the sum with `pending_scale` = 1 / world for the optimizer kernel to apply (no extra pass over 81 MB)."""
        self.armed = False
        self.pending_scale = 1.0
        if not self.active():
            for bk in self.buckets:
                bk.count, bk.work = 0, None
            return
        world = dist.get_world_size(self.group)
        for bk in self.buckets:
            if bk.work is not None and bk.count != bk.expected:           # cannot happen (the hook raises); belt and braces
                raise RuntimeError('OverlapReducer: %s bucket launched after %d of %d accumulations'
                                   % (bk.name, bk.count, bk.expected))
        # tail region(s): asynchronously as well, so that all regions' transfers are in flight together
        f = self.fg.flat
        late = [(self._async(f[lo:lo + n]), lo, n) for lo, n in self.rest if n]
        for bk in self.buckets:
            if bk.work is None:                   # never launched early (undeclared step, empty batches, fewer passes)
                late.append((self._async(bk.view), bk.lo, bk.n))
            else:
                self._wait(bk)
        for w, lo, n in late:
            self._after(w, lo, n)
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        self.partials_ready = self._done_regions == set(r for r in self.regions() if r[1]) and \
            (self.sumsq_fn is not None or self.stream is None)
        if self.average:
            if fold_scale:
                self.pending_scale = 1.0 / world
            else:
                self.fg.flat.div_(world)
                self.bucket_sumsq = {k: v / float(world) ** 2 for k, v in self.bucket_sumsq.items()}
        for bk in self.buckets:
            bk.count, bk.work = 0, None

    def total_norm(self):
        """sqrt of the summed per-region sums of squares of the last finished step (CPU tensors)."""
        return float(sum(self.bucket_sumsq.values())) ** 0.5


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
        # the score head's parameters FIRST in both flat buffers, the two encoders' right behind them: the early bucket is
        # [0, n), the middle bucket [n, n + m) and everything else ONE contiguous tail -- three collectives per step
        names = dict((id(p), n) for n, p in module.named_parameters())
        head = [p for n, p in module.named_parameters() if n in ('linear.weight', 'linear.bias')]
        mid = [p for n, p in module.named_parameters() if n.startswith(('encoder.', 'encoder_r.'))] if head else []
        if os.environ.get('RENET_REDUCER_BUCKETS', '3') == '2':         # (the two-bucket exchange of rounds 2-5, for A/B runs)
            mid = []
        self.params = FlatParams(module, first=head + mid)
        self.grads = FlatGrads(module, first=head + mid)     # same layout (flat_layout) as the parameters
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
        if head:                                                      # RENet: overlap the score head's / the encoders' buckets
            mid_names = tuple(names[id(p)] for p in mid)
            self._part = torch.zeros(self.PART_SLOTS * 4, device=self.m.device, dtype=torch.float32)
            self.reducer = OverlapReducer(self.grads, self.grads.span(('linear.weight', 'linear.bias'), module), head,
                                          mid_span=self.grads.span(mid_names, module) if mid else None, mid_params=mid,
                                          sumsq_fn=self._region_sumsq if self.m.is_cuda else None)
            # registry keyed by parameter identity (a second optimizer must not steal a global hook);
            # close() / garbage collection of this optimizer unregisters
            import ops
            self._hook = ops.register_grad_done_hook(head + mid, self.reducer.on_grad_done,
                                                     wanted=lambda r=self.reducer: r.armed and r.active())

    PART_SLOTS = 512          # sum-of-squares partials per region of the exchange (at most 4 regions: 2048 workgroup slots)

    def _region_sumsq(self, k, region):
        """Called by the reducer on ITS stream right behind region k's all-reduce: that region's partial sums of g^2."""
        self.K.sumsq_partials(region, self._part, k * self.PART_SLOTS, self.PART_SLOTS)

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
        pre = None
        if self.reducer is not None and self.reducer.partials_ready and self.reducer.sumsq_fn is not None:
            # every region's sum of squares was accumulated behind its all-reduce: only the scalar combine is left
            pre = (self._part, self.PART_SLOTS * len(self.reducer.regions()))
        self.K.adam_step(self.params.flat, self.grads.flat, self.m, self.v, self.lr, self.betas[0], self.betas[1],
                         self.eps, self.wd, self.max_norm, self.t, True, self.norm, grad_scale=scale, presummed=pre)
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
            for bk in r.buckets:
                if bk.work is not None:
                    r._wait(bk)
                bk.count = 0
        return False
