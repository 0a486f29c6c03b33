#!/usr/bin/env python
"""Host-side cost of the data-parallel exchange on ONE GPU: bench.py's step with a one-rank RCCL group
(RENET_FORCE_REDUCER=1) against the plain step -- host milliseconds per phase (no synchronisation inside the loop) and
the wall time per step.     python tools/dp_overhead.py [steps]"""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 're-net_amd'))


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    forced = os.environ.get('RENET_FORCE_REDUCER') == '1'
    dev = torch.device('cuda:0')
    torch.cuda.set_device(0)
    if forced:
        for k, v in (('MASTER_ADDR', '127.0.0.1'), ('MASTER_PORT', '29519'), ('RANK', '0'), ('WORLD_SIZE', '1')):
            os.environ.setdefault(k, v)
        dist.init_process_group('nccl', device_id=dev)
    import model as M
    import parallel
    import preprocess as P
    import synth
    quads, num_ent, num_rels, _ = synth.make_stream('ICEWS18', seed=999)
    gd = P.build_graph_dict(quads, num_rels)
    hs, ho = P.HistoryIndex(quads, 's', 10), P.HistoryIndex(quads, 'o', 10)
    torch.manual_seed(999)
    net = M.RENet(num_ent, 200, num_rels, dropout=0.5, seq_len=10, num_k=10)
    gen = torch.Generator().manual_seed(7)
    net.global_emb = {int(t): torch.randn(1, 1, 200, generator=gen) * 0.1 for t in gd}
    net.to(dev).train()
    opt = parallel.HipAdam(net, lr=1e-3, weight_decay=1e-5, max_norm=1.0)
    perm = np.random.RandomState(999).permutation(len(quads))
    preps = []
    for k in range(8):
        idx = perm[k * 1024:(k + 1) * 1024]
        preps.append(net.prepare_both(quads[idx], hs.take(idx), ho.take(idx), gd))
    t = {'fwd': 0.0, 'bwd': 0.0, 'step': 0.0}
    for it in range(steps + 5):
        if it == 5:
            torch.cuda.synchronize()
            t = {k: 0.0 for k in t}
            w0 = time.perf_counter()
        with opt.step_scope(head_passes=1):
            a = time.perf_counter()
            loss = net.loss_prepared_both(preps[it % 8])
            b = time.perf_counter()
            loss.backward()
            c = time.perf_counter()
            opt.step()
            d = time.perf_counter()
        t['fwd'] += b - a
        t['bwd'] += c - b
        t['step'] += d - c
    torch.cuda.synchronize()
    wall = (time.perf_counter() - w0) / steps
    print('forced one-rank RCCL' if forced else 'plain', ': wall %.3f ms per step; host ms per step: fwd %.3f bwd %.3f opt.step %.3f'
          % (wall * 1e3, t['fwd'] / steps * 1e3, t['bwd'] / steps * 1e3, t['step'] / steps * 1e3))
    if forced:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
