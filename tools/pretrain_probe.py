#!/usr/bin/env python
"""Epoch losses of the global model's pretraining loop (pretrain.py:60-96; batch 1024 = one step per epoch, lr 1e-2, 3 epochs) on all of YAGO
on the HIP path, per seed and dropout -- to be read against the reference's own prints in the fixture generator's log
(tools/make_e2e_full_golden.py).   python tools/pretrain_probe.py <dropout> <seed> [seed ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 're-net_amd')):
    sys.path.insert(0, p)


def main():
    from sklearn.utils import shuffle
    import global_model as GM
    import ops
    import renet_hip as K
    import utils as U
    K.lib()
    drop = float(sys.argv[1])
    d = np.load(os.path.join(ROOT, 'tests', 'golden', 'yago_full.npz'))
    tr = d['train'].astype(np.int64)
    ne, nr = int(d['num_ent']), int(d['num_rels'])
    gd = U.build_graph_dict(tr, nr)
    times = np.unique(tr[:, 3])
    tp_s, tp_o = U.get_true_distribution(tr, ne)
    dev = torch.device('cuda:0')
    for seed in [int(x) for x in sys.argv[2:]]:
        np.random.seed(seed)
        torch.manual_seed(seed)
        ops.reset_seed_counter()
        gnet = GM.RENet_global(ne, 200, nr, dropout=drop, model=0, seq_len=10, num_k=1000, maxpool=1).to(dev)
        opt = torch.optim.Adam(gnet.parameters(), lr=1e-2, weight_decay=1e-5)
        losses = []
        for ep in range(3):
            gnet.train()
            tt, ps, po = shuffle(times, tp_s, tp_o)
            tot = 0.0
            for bt, bs, bo in U.make_batch(tt, ps, po, 1024):
                loss = gnet(torch.from_numpy(bt), torch.from_numpy(bs), torch.from_numpy(bo), gd)
                loss.backward()
                torch.nn.utils.clip_grad_norm_(gnet.parameters(), 1.0)
                opt.step()
                opt.zero_grad()
                tot += loss.item()
            losses.append(tot / (len(times) / 1024))
        print('dropout %.1f seed %d HIP pretrain epoch losses %s' % (drop, seed, np.round(losses, 5).tolist()), flush=True)


if __name__ == '__main__':
    main()
