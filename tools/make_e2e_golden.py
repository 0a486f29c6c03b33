#!/usr/bin/env python
"""End-to-end parity fixture on REAL data: a prefix of YAGO (data/YAGO of the reference tree) is
trained with the UNMODIFIED reference modules (model.RENet, global_model.RENet_global, utils under
oracle/dgl_shim.py, CPU) following train.py's loop (train.py:29-31 seeds, :127-143 step, :151-185
filtered validation), and the per-epoch losses + filtered validation ranks are stored.  The GPU test
tests/test_gpu_e2e.py re-runs the same loop on the HIP path from the identical initial state (same torch
seed => same parameter init, same shuffles, recorded entity samples) and compares.

Writes tests/golden/yago_prefix.npz (quadruples: public YAGO data, ~30 timestamps) and
tests/golden/e2e_yago.npz (results).   python tools/make_e2e_golden.py [n_train_t n_valid_t epochs]
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
CFG = dict(h=200, seq_len=10, batch=1024, num_k=50, lr=1e-3, wd=1e-5, grad_norm=1.0, dropout=0.0, maxpool=1)


def load_yago(n_train_t, n_valid_t):
    ref = ref_loader.load()
    d = os.path.join(ref_loader.REFERENCE_ROOT, 'data', 'YAGO')
    num_ent, num_rels = ref.utils.get_total_number(d, 'stat.txt')
    train, times = ref.utils.load_quadruples(d, 'train.txt')
    t_tr = times[:n_train_t]
    t_va = times[n_train_t:n_train_t + n_valid_t]
    t_te = times[n_train_t + n_valid_t:n_train_t + 2 * n_valid_t]
    pick = lambda ts: train[np.isin(train[:, 3], ts)]
    return num_ent, num_rels, pick(t_tr), pick(t_va), pick(t_te)


def main():
    n_train_t = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    n_valid_t = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    tag = sys.argv[4] if len(sys.argv) > 4 else ''
    ref = ref_loader.load()
    from sklearn.utils import shuffle
    num_ent, num_rels, tr, va, te = load_yago(n_train_t, n_valid_t)
    np.savez_compressed(os.path.join(OUT, 'yago_prefix%s.npz' % tag), train=tr, valid=va, test=te,
                        num_ent=num_ent, num_rels=num_rels)
    print('YAGO prefix: train %d valid %d test %d quads' % (len(tr), len(va), len(te)), flush=True)
    # histories exactly as data/YAGO/get_history_graph.py builds them (restated + pinned in oracle tests)
    (sh, sht), (oh, oht), st = O.build_histories(tr, num_ent)
    (vsh, vsht), (voh, voht), st = O.build_histories(va, num_ent, state=st)
    (tsh, tsht), (toh, toht), st = O.build_histories(te, num_ent, state=st)
    with ref_loader.cpu_mode():
        graph_dict = {}
        for t in np.unique(tr[:, 3]):
            graph_dict[t] = ref.utils.get_big_graph(tr[tr[:, 3] == t][:, :3], num_rels)
        seed = 999                                                       # train.py:29-31
        np.random.seed(seed)
        torch.manual_seed(seed)
        model = ref.model.RENet(num_ent, CFG['h'], num_rels, dropout=CFG['dropout'], model=0,
                                seq_len=CFG['seq_len'], num_k=CFG['num_k'])
        gmodel = ref.global_model.RENet_global(num_ent, CFG['h'], num_rels, dropout=CFG['dropout'], model=0,
                                               seq_len=CFG['seq_len'], num_k=CFG['num_k'], maxpool=CFG['maxpool'])
        opt = torch.optim.Adam(model.parameters(), lr=CFG['lr'], weight_decay=CFG['wd'])
        with torch.no_grad():
            model.global_emb = gmodel.get_global_emb(np.unique(tr[:, 3]), graph_dict)
        model.graph_dict = graph_dict
        losses, step_losses = [], []
        for ep in range(epochs):
            model.train()
            t0 = time.time()
            d_, a, b, c, d2 = shuffle(tr, sh, sht, oh, oht)              # train.py:127
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
                step_losses.append(loss.item())
            losses.append(tot / (len(tr) / CFG['batch']))
            print('epoch %d loss %.6f (%.0f s)' % (ep + 1, losses[-1], time.time() - t0), flush=True)
        # filtered validation (train.py:151-185) with the random entity samples recorded
        model.eval(); gmodel.eval()
        samples = []
        Cat = torch.distributions.categorical.Categorical
        orig = Cat.sample

        def rec(self, shape=torch.Size()):
            out = orig(self, shape)
            samples.append(out.clone())
            return out
        Cat.sample = rec
        ranks, vlosses = [], []
        try:
            with torch.no_grad():
                total = torch.from_numpy(np.concatenate((tr, va, te)))
                valid = torch.from_numpy(va)
                model.init_history(tr, (sh, sht), (oh, oht), valid, (vsh, vsht), (voh, voht), te, (tsh, tsht),
                                   (toh, toht))
                model.latest_time = valid[0][3]
                t0 = time.time()
                for i in range(len(va)):
                    rk, l = model.evaluate_filter(valid[i], (vsh[i], vsht[i]), (voh[i], voht[i]), gmodel, total)
                    ranks.append(rk)
                    vlosses.append(l.item())
                print('validation %.0f s' % (time.time() - t0), flush=True)
        finally:
            Cat.sample = orig
    ranks = np.asarray(ranks)
    m = O.mrr_hits(ranks.reshape(-1))
    print('reference: filtered MRR %.6f hits@1/3/10 %.4f %.4f %.4f' % (m['mrr'], m['hits@1'], m['hits@3'], m['hits@10']))
    np.savez_compressed(os.path.join(OUT, 'e2e_yago%s.npz' % tag), epoch_loss=np.asarray(losses),
                        step_loss=np.asarray(step_losses), ranks=ranks, valid_loss=np.asarray(vlosses),
                        samples=np.stack([x.numpy() for x in samples]) if samples else np.zeros((0, CFG['num_k'])),
                        epochs=epochs, **{k: np.asarray(v) for k, v in CFG.items()})


if __name__ == '__main__':
    main()
