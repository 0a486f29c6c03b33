#!/usr/bin/env python
"""Where the launching thread spends the ~2.1 ms it needs to ENQUEUE one merged training step (bench.py's `host_enqueue_ms_per_step`):
cProfile over 60 steps of the product loop on device-resident batches (no synchronisation inside the profiled region).
    python tools/host_profile.py [top=45]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 're-net_amd'))


def main():
    top = int(sys.argv[1]) if len(sys.argv) > 1 else 45
    import renet_hip as K
    K.lib()
    import model as M
    import parallel
    import preprocess as P
    import synth
    dev = torch.device('cuda:0')
    quads, ne, nr, _ = synth.make_stream('ICEWS18', seed=999)
    gd = P.build_graph_dict(quads, nr)
    hs, ho = P.HistoryIndex(quads, 's', 10), P.HistoryIndex(quads, 'o', 10)
    torch.manual_seed(999)
    net = M.RENet(ne, 200, nr, dropout=0.5, seq_len=10, num_k=1000)
    gen = torch.Generator().manual_seed(7)
    net.global_emb = {int(t): torch.randn(1, 1, 200, generator=gen) * 0.1 for t in gd}
    net.to(dev).train()
    opt = parallel.HipAdam(net, lr=1e-3, weight_decay=1e-5, max_norm=1.0)
    perm = np.random.RandomState(999).permutation(len(quads))
    n = 70
    preps = []
    for k in range(n):
        idx = parallel.shard_indices(perm, k, 0, 1, 1024)
        preps.append(net.prepare_both(quads[idx], hs.take(idx), ho.take(idx), gd))

    def step(p):
        with opt.step_scope(head_passes=1):
            loss = net.loss_prepared_both(p)
            loss.backward()
            opt.step()
    for k in range(10):
        step(preps[k])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(10, 40):
        step(preps[k])
    enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print('unprofiled: enqueue %.3f ms per step, wall %.3f ms per step' % (enq * 1e3 / 30, wall * 1e3 / 30))
    pr = cProfile.Profile()
    pr.enable()
    for k in range(40, 70):
        step(preps[k])
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats('tottime').print_stats(top)
    st.sort_stats('cumulative').print_stats(30)


if __name__ == '__main__':
    main()
