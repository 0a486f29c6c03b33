#!/usr/bin/env python
"""FULL-LENGTH reference run (round 5): the UNMODIFIED reference modules (oracle/ref_loader.py under oracle/dgl_shim.py) on CPU over
ALL of public YAGO with the README's schedule (README.md:57-69): global model pretrained 20 epochs at lr 1e-3 (batch 1024 = one step
per epoch), RE-Net trained `epochs` (default 20) epochs at lr 1e-3 / batch 1024 / dropout 0.5 / num_k 1000, then train.py's filtered
validation (train.py:151-185) on the final epoch and test.py's loop (test.py:96-139) over the test split from the state the validation
pass leaves behind (what train.py's checkpoint holds).  ~15-19 minutes of CPU per epoch here: one seed is what a round affords.

    python tools/make_full_length_reference.py [seed=999] [epochs=20]      -> tests/golden/e2e_yago_full_len<epochs>.npz

Every epoch's mean loss is written to the log and to <out>.partial.npz as it completes (a run cut short still leaves its loss curve).
The HIP counterpart of the same schedule: RENET_FULL_TEST=1 RENET_FULL_PRE_LR=1e-3 tools/yago_full_run.py 0.5 <epochs> 20 <seed>."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from oracle import ref_loader, renet_oracle as O   # noqa: E402
import make_e2e_full_golden as F                   # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
CFG = dict(h=200, seq_len=10, batch=1024, num_k=1000, lr=1e-3, wd=1e-5, grad_norm=1.0, dropout=0.5, maxpool=1, pre_epochs=20,
           pre_batch=1024, pre_lr=1e-3)


def main():
    from sklearn.utils import shuffle
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 999
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    torch.set_num_threads(int(os.environ.get('RENET_GOLDEN_THREADS', '8')))
    out = os.path.join(OUT, 'e2e_yago_full_len%d.npz' % epochs)
    ref = ref_loader.load()
    tr, va, te, num_ent, num_rels = F.load_full()
    (sh, sht), (oh, oht), st = O.build_histories(tr, num_ent)
    (vsh, vsht), (voh, voht), st = O.build_histories(va, num_ent, state=st)
    (tsh, tsht), (toh, toht), st = O.build_histories(te, num_ent, state=st)
    graph_dict = {t: ref.utils.get_big_graph(tr[tr[:, 3] == t][:, :3], num_rels) for t in np.unique(tr[:, 3])}
    times = np.unique(tr[:, 3])
    with ref_loader.cpu_mode():
        np.random.seed(seed)
        torch.manual_seed(seed)
        gmodel = ref.global_model.RENet_global(num_ent, CFG['h'], num_rels, dropout=CFG['dropout'], model=0, seq_len=CFG['seq_len'],
                                               num_k=CFG['num_k'], maxpool=CFG['maxpool'])
        model = ref.model.RENet(num_ent, CFG['h'], num_rels, dropout=CFG['dropout'], model=0, seq_len=CFG['seq_len'],
                                num_k=CFG['num_k'])
        gopt = torch.optim.Adam(gmodel.parameters(), lr=CFG['pre_lr'], weight_decay=1e-5)
        tp_s, tp_o = ref.utils.get_true_distribution(tr, num_ent)
        pre_losses = []
        for ep in range(CFG['pre_epochs']):                                    # pretrain.py:60-96
            gmodel.train()
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
        print('pretrain losses', np.round(pre_losses, 3).tolist(), flush=True)
        gmodel.eval()
        with torch.no_grad():
            gmodel.global_emb = gmodel.get_global_emb(times, graph_dict)
        model.global_emb = gmodel.global_emb
        model.graph_dict = graph_dict
        opt = torch.optim.Adam(model.parameters(), lr=CFG['lr'], weight_decay=CFG['wd'])
        losses = []
        for ep in range(epochs):                                               # train.py:118-143
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
            print('seed %d epoch %d loss %.5f (%.0f s)' % (seed, ep + 1, losses[-1], time.time() - t0), flush=True)
            np.savez_compressed(out.replace('.npz', '.partial.npz'), seed=seed, epoch_loss=np.asarray(losses),
                                pre_loss=np.asarray(pre_losses), **{k: np.asarray(v) for k, v in CFG.items()})
        model.eval()
        gmodel.eval()
        with torch.no_grad():
            total = torch.from_numpy(np.concatenate((tr, va, te)))
            valid, test = torch.from_numpy(va), torch.from_numpy(te)
            model.init_history(tr, (sh, sht), (oh, oht), valid, (vsh, vsht), (voh, voht), te, (tsh, tsht), (toh, toht))
            model.latest_time = valid[0][3]
            t0 = time.time()
            vr = [model.evaluate_filter(valid[i], (vsh[i], vsht[i]), (voh[i], voht[i]), gmodel, total)[0] for i in range(len(va))]
            print('validation %.0f s' % (time.time() - t0), flush=True)
            for ee in range(num_ent):                                          # test.py:96-103
                while len(model.s_hist_test[ee]) > CFG['seq_len']:
                    model.s_hist_test[ee].pop(0)
                    model.s_hist_test_t[ee].pop(0)
                while len(model.o_hist_test[ee]) > CFG['seq_len']:
                    model.o_hist_test[ee].pop(0)
                    model.o_hist_test_t[ee].pop(0)
            t0 = time.time()
            trk = [model.evaluate_filter(test[i], (tsh[i], tsht[i]), (toh[i], toht[i]), gmodel, total)[0] for i in range(len(te))]
            print('test %.0f s' % (time.time() - t0), flush=True)
    vr, trk = np.asarray(vr), np.asarray(trk)
    mv, mt = O.mrr_hits(vr.reshape(-1)), O.mrr_hits(trk.reshape(-1))
    print('seed %d, %d epochs: valid MRR %.6f | TEST MRR %.6f hits@1/3/10 %.4f %.4f %.4f' % (
        seed, epochs, mv['mrr'], mt['mrr'], mt['hits@1'], mt['hits@3'], mt['hits@10']), flush=True)
    np.savez_compressed(out, seed=seed, epochs=epochs, epoch_loss=np.asarray(losses), pre_loss=np.asarray(pre_losses),
                        valid_mrr=mv['mrr'], valid_hits=np.asarray([mv['hits@1'], mv['hits@3'], mv['hits@10']]),
                        test_mrr=mt['mrr'], test_hits=np.asarray([mt['hits@1'], mt['hits@3'], mt['hits@10']]),
                        valid_ranks=vr.astype(np.int32), test_ranks=trk.astype(np.int32),
                        **{k: np.asarray(v) for k, v in CFG.items()})


if __name__ == '__main__':
    main()
