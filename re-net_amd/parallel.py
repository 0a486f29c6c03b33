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
(ops.grad_done_hook) and runs under the GRU / RGCN backward (~1.5 ms); the rest follows in step().
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


class FlatGrads(object):
    """Makes every parameter's .grad a view into one contiguous buffer."""

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]
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
    entity score head, model.py:38, whose backward runs first); `early_uses` in-place accumulations complete it
    per step (the subject and the object pass: 2).  on_grad_done(param) is called by the autograd Functions right
    after they accumulated into param.grad (ops.grad_done_hook); when the early bucket is complete its all-reduce
    is issued asynchronously on a side stream (device tensors) / as an async gloo op (CPU tensors) and proceeds
    while the rest of the backward pass runs.  finish() waits for it, reduces the remaining bucket and averages."""

    def __init__(self, flat_grads, early_span, early_params, early_uses=2, group=None):
        self.fg, self.group = flat_grads, group
        self.lo, self.n = early_span
        self.early_ids = {id(p) for p in early_params}
        self.early_uses = early_uses * len(self.early_ids)
        self.count, self.work = 0, None
        self.average = True                       # False: SUM (ranks hold disjoint shares of one batch, bench --scaling exact)
        f = flat_grads.flat
        self.early = f[self.lo:self.lo + self.n]
        self.rest = [f[:self.lo], f[self.lo + self.n:]]
        self.stream = torch.cuda.Stream() if f.is_cuda else None

    def set_uses(self, n):
        """in-place accumulations that complete the early bucket per step: 2 for the subject + object passes,
        1 when a step runs them as one merged pass (RENet.loss_prepared_both)"""
        self.early_uses = n * len(self.early_ids)

    def active(self):
        # RENET_FORCE_REDUCER=1 runs the collectives even in a one-rank group (a no-op exchange): lets a one-GPU box
        # exercise the side-stream / RCCL call sequence
        if not (dist.is_available() and dist.is_initialized()):
            return False
        return dist.get_world_size(self.group) > 1 or os.environ.get('RENET_FORCE_REDUCER') == '1'

    def on_grad_done(self, param):
        if id(param) not in self.early_ids or not self.active():
            return
        self.count += 1
        if self.count == self.early_uses and self.work is None:
            if self.stream is not None:
                self.stream.wait_stream(torch.cuda.current_stream())      # the accumulating kernels are queued
                with torch.cuda.stream(self.stream):
                    self.work = dist.all_reduce(self.early, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            else:
                self.work = dist.all_reduce(self.early, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        """Call after backward(): completes the exchange; the flat buffer then holds the rank-averaged gradient."""
        if not self.active():
            self.count, self.work = 0, None
            return
        world = dist.get_world_size(self.group)
        for t in self.rest:
            if t.numel():
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        if self.work is None:                     # the early bucket never completed early (e.g. empty batches)
            dist.all_reduce(self.early, op=dist.ReduceOp.SUM, group=self.group)
        else:
            self.work.wait()                      # device: makes the current stream wait for the side stream
            if self.stream is not None:
                torch.cuda.current_stream().wait_stream(self.stream)
        if self.average:
            self.fg.flat.div_(world)
        self.count, self.work = 0, None


def shard_indices(perm, step, rank, world, batch_size):
    """Rank `rank`'s quadruple indices for global step `step`: consecutive batch_size slices of the
    shuffled order are dealt round-robin to the ranks (weak scaling: per-rank batch is fixed)."""
    k = step * world + rank
    n = len(perm)
    start = (k * batch_size) % max(n - batch_size + 1, 1)
    return perm[start:start + batch_size]


class FlatParams(object):
    """Moves every parameter of `module` into ONE contiguous buffer (param.data become views), in the same
    order as FlatGrads, so that the optimizer is a single fused kernel over flat buffers."""

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]
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
    the flat buffers (renet_adam_step).  Matches train.py:61,140-142."""

    def __init__(self, module, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_norm=0.0):
        import renet_hip as K
        self.K = K
        self.params = FlatParams(module)
        self.grads = FlatGrads(module)            # same layout (flat_layout) as the parameters
        self.m = torch.zeros_like(self.params.flat)
        self.v = torch.zeros_like(self.params.flat)
        self.lr, self.betas, self.eps, self.wd, self.max_norm = lr, betas, eps, weight_decay, max_norm
        self.t = 0
        self.norm = torch.zeros(1, device=self.m.device, dtype=torch.float32)
        self.reducer = None
        names = [n for n, _ in module.named_parameters()]
        if 'linear.weight' in names and 'linear.bias' in names:       # RENet: overlap the score head's bucket
            import ops
            early = [p for n, p in module.named_parameters() if n in ('linear.weight', 'linear.bias')]
            self.reducer = OverlapReducer(self.grads, self.grads.span(('linear.weight', 'linear.bias'), module), early)
            ops.grad_done_hook = self.reducer.on_grad_done

    def step(self):
        """all-reduce (if distributed; the score head's bucket was started during backward) -> clip -> Adam ->
        zero_grad."""
        if self.reducer is not None:
            self.reducer.finish()
        else:
            self.grads.allreduce_mean()
        self.t += 1
        self.K.adam_step(self.params.flat, self.grads.flat, self.m, self.v, self.lr, self.betas[0], self.betas[1],
                         self.eps, self.wd, self.max_norm, self.t, True, self.norm)
