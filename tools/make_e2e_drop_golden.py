#!/usr/bin/env python
"""TRAIN-MODE accuracy fixture (round 4): the mode bench.py times -- dropout 0.5 at all four sites
(RGCN.py:36-37, Aggregator.py:157-158, model.py:90,99) -- run with the UNMODIFIED reference modules on CPU
(oracle/ref_loader.py under oracle/dgl_shim.py), several seeds:

  per seed:  pretrain.py's loop for the global model (pretrain.py:60-96; 2 epochs, dropout 0.5)
             -> get_global_emb -> train.py's loop (train.py:127-143; `epochs` epochs, dropout 0.5)
             -> train.py's filtered validation (train.py:151-185) -> MRR / Hits@1,3,10.

Dropout masks come from torch's CPU generator, the HIP path's from its own counters: the comparison is
statistical (tests/test_gpu_e2e.py::test_train_mode_filtered_mrr_matches_reference_statistically compares the
per-seed means of this file with the HIP path's per-seed means, per GEMM mode).

Data: tests/golden/yago_prefix_big.npz (41 452 / 2 799 / 3 005 quadruples of public YAGO).
Writes tests/golden/e2e_yago_drop.npz.      python tools/make_e2e_drop_golden.py [epochs] [seed ...]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_loader, renet_oracle as O   # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
CFG = dict(h=200, seq_len=10, batch=1024, num_k=50, lr=1e-3, wd=1e-5, grad_norm=1.0, dropout=0.5, maxpool=1,
           pre_epochs=2, pre_batch=16, pre_lr=1e-3)


def run_seed(ref, seed, epochs, tr, va, te, num_ent, num_rels, hist):
    from sklearn.utils import shuffle
    (sh, sht), (oh, oht), (vsh, vsht), (voh, voht), (tsh, tsht), (toh, toht) = hist
    graph_dict = {}
    for t in np.unique(tr[:, 3]):
        graph_dict[t] = ref.utils.get_big_graph(tr[tr[:, 3] == t][:, :3], num_rels)
    times = np.unique(tr[:, 3])
    np.random.seed(seed)                                                 # train.py:29-31 / pretrain.py:19-21
    torch.manual_seed(seed)
    gmodel = ref.global_model.RENet_global(num_ent, CFG['h'], num_rels, dropout=CFG['dropout'], model=0,
                                           seq_len=CFG['seq_len'], num_k=CFG['num_k'], maxpool=CFG['maxpool'])
    model = ref.model.RENet(num_ent, CFG['h'], num_rels, dropout=CFG['dropout'], model=0,
                            seq_len=CFG['seq_len'], num_k=CFG['num_k'])
    # ---- pretrain.py:60-96
    gopt = torch.optim.Adam(gmodel.parameters(), lr=CFG['pre_lr'], weight_decay=1e-5)
    tp_s, tp_o = ref.utils.get_true_distribution(tr, num_ent)
    pre_losses = []
    for ep in range(CFG['pre_epochs']):
        gmodel.train()
        t0 = time.time()
        tt, ps, po = shuffle(times, tp_s, tp_o)
        tot = 0.0
        for bt, bs, bo in ref.utils.make_batch(tt, ps, po, CFG['pre_batch']):
            loss = gmodel(torch.from_numpy(bt), torch.from_numpy(bs), torch.from_numpy(bo), graph_dict)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(gmodel.parameters(), CFG['grad_norm'])
            gopt.step()
            gopt.zero_grad()
            tot += loss.item()
        pre_losses.append(tot / (len(times) / CFG['pre_batch']))
        print('  seed %d pretrain epoch %d loss %.5f (%.0f s)' % (seed, ep + 1, pre_losses[-1], time.time() - t0),
              flush=True)
    gmodel.eval()
    with torch.no_grad():
        gmodel.global_emb = gmodel.get_global_emb(times, graph_dict)     # pretrain.py:91
    model.global_emb = gmodel.global_emb                                 # train.py:79-80 (from the checkpoint)
    model.graph_dict = graph_dict
    # ---- train.py:118-143
    opt = torch.optim.Adam(model.parameters(), lr=CFG['lr'], weight_decay=CFG['wd'])
    losses = []
    for ep in range(epochs):
        model.train()
        t0 = time.time()
        d_, a, b, c, d2 = shuffle(tr, sh, sht, oh, oht)
        tot = 0.0
        for bd, bs, bst, bo, bot in ref.utils.make_batch2(d_, a, b, c, d2, CFG['batch']):
            bd = torch.from_numpy(bd).long()
            loss = model(bd, (bs, bst), (bo, bot), graph_dict, subject=True) + \
                model(bd, (bs, bst), (bo, bot), graph_dict, subject=False)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), CFG['grad_norm'])
            opt.step()
            opt.zero_grad()
            tot += loss.item()
        losses.append(tot / (len(tr) / CFG['batch']))
        print('  seed %d epoch %d loss %.5f (%.0f s)' % (seed, ep + 1, losses[-1], time.time() - t0), flush=True)
    # ---- train.py:151-185
    model.eval()
    gmodel.eval()
    ranks = []
    with torch.no_grad():
        total = torch.from_numpy(np.concatenate((tr, va, te)))
        valid = torch.from_numpy(va)
        model.init_history(tr, (sh, sht), (oh, oht), valid, (vsh, vsht), (voh, voht), te, (tsh, tsht), (toh, toht))
        model.latest_time = valid[0][3]
        t0 = time.time()
        for i in range(len(va)):
            rk, _ = model.evaluate_filter(valid[i], (vsh[i], vsht[i]), (voh[i], voht[i]), gmodel, total)
            ranks.append(rk)
        print('  seed %d validation %.0f s' % (seed, time.time() - t0), flush=True)
    ranks = np.asarray(ranks)
    m = O.mrr_hits(ranks.reshape(-1))
    print('seed %d: filtered MRR %.6f hits@1/3/10 %.4f %.4f %.4f' % (seed, m['mrr'], m['hits@1'], m['hits@3'],
                                                                     m['hits@10']), flush=True)
    return dict(pre_loss=pre_losses, epoch_loss=losses, ranks=ranks, mrr=m['mrr'],
                hits=[m['hits@1'], m['hits@3'], m['hits@10']])


def main():
    epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    seeds = [int(x) for x in sys.argv[2:]] or [999, 1000, 1001]
    ref = ref_loader.load()
    data = np.load(os.path.join(OUT, 'yago_prefix_big.npz'))
    tr, va, te = data['train'], data['valid'], data['test']
    num_ent, num_rels = int(data['num_ent']), int(data['num_rels'])
    (sh, sht), (oh, oht), st = O.build_histories(tr, num_ent)
    (vsh, vsht), (voh, voht), st = O.build_histories(va, num_ent, state=st)
    (tsh, tsht), (toh, toht), st = O.build_histories(te, num_ent, state=st)
    hist = ((sh, sht), (oh, oht), (vsh, vsht), (voh, voht), (tsh, tsht), (toh, toht))
    res = []
    with ref_loader.cpu_mode():
        for seed in seeds:
            res.append(run_seed(ref, seed, epochs, tr, va, te, num_ent, num_rels, hist))
            np.savez_compressed(
                os.path.join(OUT, 'e2e_yago_drop.npz'), seeds=np.asarray(seeds[:len(res)]), epochs=epochs,
                mrr=np.asarray([r['mrr'] for r in res]), hits=np.asarray([r['hits'] for r in res]),
                epoch_loss=np.asarray([r['epoch_loss'] for r in res]),
                pre_loss=np.asarray([r['pre_loss'] for r in res]),
                ranks=np.stack([r['ranks'] for r in res]), **{k: np.asarray(v) for k, v in CFG.items()})
    print('reference, dropout %.1f, %d seeds: MRR mean %.6f sd %.6f' % (
        CFG['dropout'], len(res), np.mean([r['mrr'] for r in res]), np.std([r['mrr'] for r in res], ddof=1)
        if len(res) > 1 else 0.0))


if __name__ == '__main__':
    main()
