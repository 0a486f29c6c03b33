#!/usr/bin/env python
"""Which torch (aten) operators still launch kernels inside one training step of bench.py's workload, and from
where: torch.profiler with stacks, grouped by (operator, first frame inside re-net_amd/).  GPU only."""
import collections
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 're-net_amd'))
import model as M
import parallel
import preprocess as P
import synth


def main():
    dev = torch.device('cuda:0')
    quads, num_ent, num_rels, _ = synth.make_stream('ICEWS18', seed=999)
    gd = P.build_graph_dict(quads, num_rels)
    hs, ho = P.HistoryIndex(quads, 's', 10), P.HistoryIndex(quads, 'o', 10)
    torch.manual_seed(999)
    net = M.RENet(num_ent, 200, num_rels, dropout=0.5, seq_len=10, num_k=1000)
    gen = torch.Generator().manual_seed(7)
    net.global_emb = {int(t): torch.randn(1, 1, 200, generator=gen) * 0.1 for t in gd}
    net.to(dev).train()
    opt = parallel.HipAdam(net, lr=1e-3, weight_decay=1e-5, max_norm=1.0)
    perm = np.random.RandomState(999).permutation(len(quads))
    preps = []
    for k in range(4):
        idx = perm[k * 1024:(k + 1) * 1024]
        preps.append((net.prepare(quads[idx], hs.take(idx), gd, True), net.prepare(quads[idx], ho.take(idx), gd, False)))

    def step(ps, po):
        loss = net.loss_prepared(ps) + net.loss_prepared(po)
        loss.backward()
        opt.step()

    for k in range(2):
        step(*preps[k])
    torch.cuda.synchronize()
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode

    groups = collections.OrderedDict()

    class Log(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            out = func(*args, **(kwargs or {}))
            name = str(func)
            if any(isinstance(a, torch.Tensor) and a.is_cuda for a in list(args) + [out] if isinstance(a, torch.Tensor)):
                frames = [f for f in traceback.extract_stack() if 're-net_amd' in f.filename or 'bench' in f.filename]
                where = '%s:%d %s' % (os.path.basename(frames[-1].filename), frames[-1].lineno, frames[-1].name) if frames else '?'
                shapes = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)]
                key = (name, where, str(shapes)[:70])
                groups[key] = groups.get(key, 0) + 1
            return out

    torch.autograd.set_multithreading_enabled(False)          # backward on this thread: the mode is thread-local
    with Log():
        step(*preps[2])
    torch.cuda.synchronize()
    skip = ('aten.view', 'aten.detach', 'aten._unsafe_view', 'aten.alias', 'aten.slice', 'aten.select', 'aten.t.',
            'aten.transpose', 'aten.expand', 'aten.as_strided', 'aten.unsqueeze', 'aten.squeeze', 'aten.reshape',
            'aten.lift_fresh', 'aten._local_scalar_dense', 'aten.empty', 'aten.permute', 'aten.unbind', 'aten.split')
    rows = [(k, n) for k, n in groups.items() if not k[0].startswith(skip)]
    print('kernel-launching aten ops in one step: %d' % sum(n for _, n in rows))
    for (name, where, shapes), n in sorted(rows, key=lambda kv: (kv[0][1], kv[0][0])):
        print('%3dx  %-34s %-44s %s' % (n, name, where, shapes))


if __name__ == '__main__':
    main()
