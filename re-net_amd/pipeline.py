"""Input pipeline: the host half of every step (batch-graph construction, graph.build_batch -- the
reference's dominant cost, utils.py:209-244) runs ahead of the device in worker PROCESSES, so the
training loop only uploads one packed buffer per direction and launches kernels.

Workers are forked (they inherit the graph store, the history index and the model's host-side tables
copy-on-write) and never touch the GPU.  Results come back in step order.
"""
import multiprocessing as mp
import os

_job = {}


def worker_budget(world=1, cpus=None, cap=24):
    """Builder worker processes PER RANK such that all ranks of a node together (world ranks x (workers + the
    training thread)) stay within the host's cores: cpus // (2 * world), at least 1, at most `cap` (one batch
    takes ~14 ms to build and ~4 ms to consume: beyond ~8 workers per rank the device is the bottleneck)."""
    cpus = cpus or os.cpu_count() or 2
    return max(1, min(int(cap), cpus // (2 * max(int(world), 1))))


def _run(step):
    return _job['fn'](step)


class BatchPrefetcher(object):
    """for host_batches in BatchPrefetcher(fn, steps, workers): ...   where fn(step) is pure host work."""

    def __init__(self, fn, steps, num_workers=None, chunksize=1, threads=False):
        self.fn, self.steps = fn, list(steps)
        self.num_workers = num_workers or max(1, min(32, (os.cpu_count() or 2) // 2))
        self.chunksize = chunksize
        self.threads = threads
        self.pool = None

    def _iter_threads(self):
        """Worker THREADS of this process instead of forked processes: the builder spends most of its time in the
        native passes (ctypes releases the GIL) and in large numpy kernels (which release it too), so a few threads
        build several future batches concurrently without copying anything between processes."""
        from concurrent.futures import ThreadPoolExecutor
        ahead = 2 * self.num_workers
        with ThreadPoolExecutor(self.num_workers) as ex:
            pending = []
            it = iter(self.steps)
            for s in it:
                pending.append(ex.submit(self.fn, s))
                if len(pending) >= ahead:
                    break
            while pending:
                item = pending.pop(0).result()
                nxt = next(it, None)
                if nxt is not None:
                    pending.append(ex.submit(self.fn, nxt))
                yield item

    def __iter__(self):
        if self.num_workers <= 1:
            for s in self.steps:
                yield self.fn(s)
            return
        if self.threads:
            for item in self._iter_threads():
                yield item
            return
        _job['fn'] = self.fn                       # visible to the forked children
        ctx = mp.get_context('fork')
        self.pool = ctx.Pool(self.num_workers)
        try:
            for item in self.pool.imap(_run, self.steps, self.chunksize):
                yield item
        finally:
            self.pool.terminate()
            self.pool.join()
            self.pool = None
            _job.pop('fn', None)


class Uploader(object):
    """Uploads packed host batches on a dedicated copy stream.  A pageable-memory H2D copy issued on the
    compute stream would queue behind every kernel of the previous step and block the training thread until
    they finish; on its own (idle) stream it completes immediately, the compute stream just waits for the
    copy's event, and the training thread keeps running ahead of the GPU."""

    def __init__(self, net):
        import torch
        self.torch = torch
        self.net = net
        self.stream = torch.cuda.Stream()

    def __call__(self, hbatch):
        torch = self.torch
        main = torch.cuda.current_stream()
        with torch.cuda.stream(self.stream):
            prep = self.net.prepare_from_host(hbatch)
        main.wait_stream(self.stream)
        g = prep.g
        # allocator: every tensor allocated under the copy stream is used on the compute stream -- including the
        # scatter plans of an all-empty-history batch and a global-embedding matrix built inside this context
        ts = [prep.o_idx, prep.s_idx, prep.r_idx]
        if g is not None:
            ts += [g._buf, g.norm, getattr(g, 'glob', None)]
        for pl in (prep.plan_s, prep.plan_r):
            if pl is not None:
                ts += [pl.order, pl.seg_ptr, pl.target]
        for t in ts:
            if t is not None and torch.is_tensor(t) and t.is_cuda:
                t.record_stream(main)
        return prep
