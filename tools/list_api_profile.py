#!/usr/bin/env python
"""cProfile of the reference's loop body (train.py:133-142) over this package's RENet -- bench.py's `value_list_api` -- to see
where the host spends the step.   python tools/list_api_profile.py [top=40]"""
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
    top = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    import model as M
    import parallel
    import preprocess as P
    import synth
    dev = torch.device('cuda:0')
    quads, ne, nr, _ = synth.make_stream('ICEWS18', seed=999)
    gd = P.build_graph_dict(quads, nr)
    hs, ho = P.HistoryIndex(quads, 's', 10), P.HistoryIndex(quads, 'o', 10)
    torch.manual_seed(999)
    model = M.RENet(ne, 200, nr, dropout=0.5, seq_len=10, num_k=1000)
    gen = torch.Generator().manual_seed(3)
    model.global_emb = {int(t): torch.randn(1, 1, 200, generator=gen) * 0.1 for t in gd}
    model.to(dev)
    model.train()
    model.fuse_directions = True
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
    perm = np.random.RandomState(1).permutation(len(quads))
    batches = []
    for k in range(45):
        idx = parallel.shard_indices(perm, k, 0, 1, 1024)
        s_hist, s_hist_t = hs.to_lists(idx)
        o_hist, o_hist_t = ho.to_lists(idx)
        batches.append((quads[idx], s_hist, s_hist_t, o_hist, o_hist_t))

    def step(b):
        batch_data, s_hist, s_hist_t, o_hist, o_hist_t = b
        batch_data = torch.from_numpy(batch_data).long().cuda()
        loss_s = model(batch_data, (s_hist, s_hist_t), (o_hist, o_hist_t), gd, subject=True)
        loss_o = model(batch_data, (s_hist, s_hist_t), (o_hist, o_hist_t), gd, subject=False)
        loss = loss_s + loss_o
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        optimizer.step()
        optimizer.zero_grad()
        return loss.item()
    for b in batches[:5]:
        step(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in batches[5:25]:
        step(b)
    torch.cuda.synchronize()
    print('unprofiled: %.3f ms per step' % ((time.perf_counter() - t0) * 1e3 / 20))
    # the same with the sync moved out of the loop: how much of the step is host work the GPU does not hide
    t0 = time.perf_counter()
    pr = cProfile.Profile()
    pr.enable()
    for b in batches[25:45]:
        step(b)
    pr.disable()
    torch.cuda.synchronize()
    print('profiled:   %.3f ms per step' % ((time.perf_counter() - t0) * 1e3 / 20))
    st = pstats.Stats(pr)
    st.sort_stats('cumulative').print_stats(top)
    st.print_callers('item')
    st.sort_stats('tottime').print_stats(18)


if __name__ == '__main__':
    main()
