#!/usr/bin/env python
"""renet_segment_add2 / renet_segment_add on the bench workload's merged batch (ICEWS18-shaped, N 92 k nodes -> 23 k entities; packed rows ->
nodes; sequences -> entities / relations): us per launch and GB/s of (rows read + targets read-modify-written).  RENET_SEGADD=one selects
the single-launch kernel of rounds 3-4.   python tools/segadd_bench.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 're-net_amd'))
import graph as G            # noqa: E402
import preprocess as P       # noqa: E402
import renet_hip as K        # noqa: E402
import synth                 # noqa: E402


def main():
    K.lib()
    dev = torch.device('cuda:0')
    quads, ne, nr, _ = synth.make_stream('ICEWS18', seed=999)
    gd = P.build_graph_dict(quads, nr)
    hs, ho = P.HistoryIndex(quads, 's', 10), P.HistoryIndex(quads, 'o', 10)
    idx = np.random.RandomState(999).permutation(len(quads))[3 * 1024:4 * 1024]
    hb = G.build_batch_both(G.store_for(gd), ne, nr, quads[idx, 0], quads[idx, 1], quads[idx, 2], hs.take(idx), ho.take(idx))
    g = G.DeviceGraph(G.PackedBatch(hb), dev)
    d = 200
    cases = [('plan_node_ent  [N -> entities] x2 (segment_add2)', g.plan_node_ent, hb.N, ne, 2),
             ('plan_node_ent  [N -> entities]', g.plan_node_ent, hb.N, ne, 1),
             ('plan_subj_row  [S -> nodes]', g.plan_subj_row, hb.S, hb.N, 1),
             ('plan_s         [2B -> entities]', g.plan_s, hb.B, ne, 1),
             ('plan_r         [2B -> relations]', g.plan_r, hb.B, 2 * nr, 1)]
    for name, plan, n_src, n_dst, k in cases:
        src0, src1 = torch.randn(n_src, d, device=dev), torch.randn(n_src, d, device=dev)
        dst0, dst1 = torch.zeros(n_dst, d, device=dev), torch.zeros(n_dst, d, device=dev)
        fn = (lambda: K.segment_add2(src0, src1, plan, dst0, dst1)) if k == 2 else (lambda: K.segment_add(src0, plan, dst0))
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        nbytes = k * (n_src * d * 4 + 2 * plan.num_segments * d * 4)
        print('%-52s U %6d  %7.1f us  %6.0f GB/s  [RENET_SEGADD=%s]' % (name, plan.num_segments, us, nbytes / us / 1e3,
                                                                    os.environ.get('RENET_SEGADD', 'default')), flush=True)
        # value check against index_add in fp64
        dst0.zero_()
        K.segment_add(src0, plan, dst0)
        tgt = torch.empty(n_src, dtype=torch.long, device=dev)
        seg = torch.repeat_interleave(torch.arange(plan.num_segments, device=dev),
                                      (plan.seg_ptr[1:] - plan.seg_ptr[:-1]).long())
        tgt[plan.order.long()] = plan.target.long()[seg]
        ref = torch.zeros(n_dst, d, device=dev, dtype=torch.float64).index_add_(0, tgt, src0.double())
        err = float((dst0.double() - ref).abs().max())
        assert err < 1e-4, err


if __name__ == '__main__':
    main()
