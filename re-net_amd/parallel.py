"""Data-parallel glue: one process per GPU, torch.distributed (backend "nccl" == RCCL on ROCm) over
xGMI.  The reference has no distributed code at all; the unit that shards is the batch
of (subject, relation) histories: every rank takes its own batch of `batch_size` quadruples with the
reference's per-batch semantics, and the ONE exchange step is an all-reduce of the flat gradient
(80.9 MB fp32 at ICEWS18 sizes) before clip + Adam, i.e. exactly gradient accumulation over
world_size reference batches.

All gradients live in ONE flat fp32 buffer (param.grad are views into it), so the exchange is a
single large collective: xGMI is point-to-point (7 links x ~153 GB/s per GPU), so one big ring
all-reduce (2*(W-1)/W * 81 MB per GPU ~ 0.9 ms at W=8) beats many small ones; no bucketing/overlap
is attempted because the whole backward is only a few ms and the last-produced gradients
(ent_embeds, linear.weight) are also the largest.
"""
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

    def step(self):
        """all-reduce (if distributed) -> clip -> Adam -> zero_grad."""
        self.grads.allreduce_mean()
        self.t += 1
        self.K.adam_step(self.params.flat, self.grads.flat, self.m, self.v, self.lr, self.betas[0], self.betas[1],
                         self.eps, self.wd, self.max_norm, self.t, True, self.norm)
