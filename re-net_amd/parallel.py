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


class FlatGrads(object):
    """Makes every parameter's .grad a view into one contiguous buffer."""

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device('cpu')
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n

    def zero(self):
        self.flat.zero_()

    def check_views(self):
        """autograd accumulates in place into an existing .grad; verify nobody replaced the views."""
        base = self.flat.data_ptr()
        off = 0
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != base + 4 * off:
                return False
            off += p.numel()
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
