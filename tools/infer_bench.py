#!/usr/bin/env python
"""Filtered evaluation (test.py's loop, model.py:216-419) on a synthetic dataset-shaped stream: one
evaluate_filter call per quadruple (the reference API) vs evaluate_filter_stream (all quadruples of a timestamp
in one batch).  GPU only.   python tools/infer_bench.py [shape] [n_timestamps] [hidden]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 're-net_amd'))
import global_model as GM
import model as M
import preprocess as P
import synth
import utils as U


def setup(shape, n_eval_t, hidden, dev, seed=7, num_k=10):
    quads, ne, nr, _ = synth.make_stream(shape, seed=999, num_t=40 + n_eval_t)
    times = np.unique(quads[:, 3])
    cut = times[-n_eval_t]
    tr, te = quads[quads[:, 3] < cut], quads[quads[:, 3] >= cut]
    torch.manual_seed(seed)
    net = M.RENet(ne, hidden, nr, dropout=0.0, seq_len=10, num_k=num_k).to(dev).eval()
    gnet = GM.RENet_global(ne, hidden, nr, dropout=0.0, seq_len=10, num_k=num_k, maxpool=1).to(dev).eval()
    hs, ho = P.HistoryIndex(quads, 's', 10), P.HistoryIndex(quads, 'o', 10)
    r_tr, r_te = np.arange(len(tr)), np.arange(len(tr), len(quads))
    gd = U.build_graph_dict(tr, nr)
    with torch.no_grad():
        net.global_emb = gnet.get_global_emb(np.unique(tr[:, 3]), gd)
        net.graph_dict = gd
        tes, teo = hs.to_lists(r_te), ho.to_lists(r_te)
        net.init_history(tr, hs.to_lists(r_tr), ho.to_lists(r_tr), torch.from_numpy(te), tes, teo, te, tes, teo)
        net.latest_time = torch.from_numpy(te)[0][3]
    g = torch.Generator(device='cpu').manual_seed(seed)
    net.sample_entities = lambda prob: torch.multinomial(prob.detach().cpu(), net.num_k, replacement=True,
                                                         generator=g).to(prob.device)
    return net, gnet, te, tes, teo, torch.from_numpy(quads).to(dev)


def main():
    shape = sys.argv[1] if len(sys.argv) > 1 else 'ICEWS18'
    n_t = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    hidden = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    dev = torch.device('cuda:0')
    res = {}
    for mode in ('sequential', 'stream'):
        net, gnet, te, tes, teo, total = setup(shape, n_t, hidden, dev)
        torch.cuda.synchronize()
        t0 = time.time()
        with torch.no_grad():
            if mode == 'sequential':
                tq = torch.from_numpy(te)
                ranks = np.asarray([net.evaluate_filter(tq[i], (tes[0][i], tes[1][i]), (teo[0][i], teo[1][i]), gnet,
                                                        total)[0] for i in range(len(te))])
            else:
                ranks, _ = net.evaluate_filter_stream(te, tes, teo, gnet, total)
        torch.cuda.synchronize()
        res[mode] = (time.time() - t0, ranks)
        print('%-10s %6d quadruples over %d timestamps: %8.2f s  %9.1f quadruples/s   MRR %.5f' % (
            mode, len(te), n_t, res[mode][0], len(te) / res[mode][0], float(np.mean(1.0 / ranks))))
    a, b = res['sequential'][1], res['stream'][1]
    print('ranks equal: %.4f   max |diff| %.1f   speed-up %.1fx' % (float(np.mean(a == b)), float(np.abs(a - b).max()),
                                                                  res['sequential'][0] / res['stream'][0]))


def advance_bench(shape='ICEWS18', hidden=200, num_k=1000, n_t=3):
    """Time of ONE timestamp advance (model.py:222-328: 2 x num_k sampled entities, a [R, N_ent] joint distribution and
    its top-k for each) at test.py's default num_k = 1000, with the fused selection kernels and with the reference's
    torch op sequence (RENET_TOPK=torch)."""
    dev = torch.device('cuda:0')
    for mode in ('fused', 'torch'):
        os.environ['RENET_TOPK'] = 'torch' if mode == 'torch' else ''
        net, gnet, te, tes, teo, total = setup(shape, n_t, hidden, dev, num_k=num_k)
        ts = np.unique(te[:, 3])
        times = []
        with torch.no_grad():
            for t in ts[1:]:
                torch.cuda.synchronize()
                t0 = time.time()
                net._advance_time(torch.tensor(int(t)), gnet)
                torch.cuda.synchronize()
                times.append(time.time() - t0)
        print('advance (%s selection), num_k %d: %s s per timestamp' % (mode, num_k, ['%.2f' % x for x in times]))
    os.environ['RENET_TOPK'] = ''


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'advance':
        advance_bench(*(sys.argv[2:3] or ['ICEWS18']))
    else:
        main()
