#!/usr/bin/env python
"""The inference advance with and without relation pruning (RENet.prune_relations) on a model that has
been TRAINED for a while on the synthetic dataset-shaped stream -- pruning needs peaked p(r | s) and p(o | s, r); a random
model's distributions are flat (the config-scale fixture scores all of its 246 528 rows).

    python tools/advance_pruned_bench.py [shape=ICEWS18] [train_steps=300] [pretrain_steps=60] [num_k=1000] [hidden=200]

Per variant (unpruned, pruned): seconds per timestamp advance, rows scored / rows total per side, and whether the two
predicted graphs are identical.  Product loop for the training (merged pass, HipAdam), pretrain.py's loop for the global
model.  GPU only (set RENET_BENCH_CPU_EMULATION=1 for a functional smoke run over the torch-CPU emulation of the wrappers)."""
import copy
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 're-net_amd'), os.path.join(ROOT, 'tests'), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def build(shape, train_steps, pre_steps, num_k, hidden, dev, n_eval_t=3, batch=1024, seed=999, dropout=0.2):
    import global_model as GM
    import model as M
    import preprocess as P
    import synth
    import utils as U
    quads, ne, nr, _ = synth.make_stream(shape, seed=seed, num_t=int(os.environ.get('RENET_BENCH_NUM_T', '60')) + n_eval_t)
    times = np.unique(quads[:, 3])
    cut = times[-n_eval_t]
    tr, te = quads[quads[:, 3] < cut], quads[quads[:, 3] >= cut]
    np.random.seed(seed)
    torch.manual_seed(seed)
    gnet = GM.RENet_global(ne, hidden, nr, dropout=dropout, seq_len=10, num_k=num_k, maxpool=1).to(dev)
    net = M.RENet(ne, hidden, nr, dropout=dropout, seq_len=10, num_k=num_k).to(dev)
    hs, ho = P.HistoryIndex(quads, 's', 10), P.HistoryIndex(quads, 'o', 10)
    r_tr, r_te = np.arange(len(tr)), np.arange(len(tr), len(quads))
    gd = U.build_graph_dict(tr, nr)
    tt = np.unique(tr[:, 3])
    # ---- pretrain.py:60-96
    t0 = time.time()
    gopt = torch.optim.Adam(gnet.parameters(), lr=1e-3, weight_decay=1e-5)
    tp_s, tp_o = U.get_true_distribution(tr, ne)
    gnet.train()
    done = 0
    while done < pre_steps:
        order = np.random.permutation(len(tt))
        for bt, bs, bo in U.make_batch(tt[order], tp_s[order], tp_o[order], 64):
            loss = gnet(torch.from_numpy(bt), torch.from_numpy(bs), torch.from_numpy(bo), gd)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(gnet.parameters(), 1.0)
            gopt.step()
            gopt.zero_grad()
            done += 1
            if done >= pre_steps:
                break
    gnet.eval()
    with torch.no_grad():
        net.global_emb = gnet.get_global_emb(tt, gd)
    net.graph_dict = gd
    # ---- train.py:118-143 as the product runs it
    import parallel
    sh, sht = hs.to_lists(r_tr)
    oh, oht = ho.to_lists(r_tr)
    hip_opt = dev.type == 'cuda'
    opt = parallel.HipAdam(net, lr=1e-3, weight_decay=1e-5, max_norm=1.0) if hip_opt else \
        torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=1e-5)          # (emulation smoke run: torch's Adam)
    net.train()
    done, last = 0, float('nan')
    while done < train_steps:
        order = np.random.permutation(len(tr))
        d_ = tr[order]
        a, b, c, d2 = [sh[i] for i in order], [sht[i] for i in order], [oh[i] for i in order], [oht[i] for i in order]
        for bd, bs, bst, bo, bot in U.make_batch2(d_, a, b, c, d2, batch):
            prep = net.prepare_both(bd, (bs, bst), (bo, bot), gd)
            loss = net.loss_prepared_both(prep)
            loss.backward()
            if not hip_opt:
                torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
            opt.step()
            if not hip_opt:
                opt.zero_grad()
            done += 1
            if done % 50 == 0 or done == train_steps:
                last = float(loss.item())
                print('  train step %d loss %.4f (%.0f s)' % (done, last, time.time() - t0), flush=True)
            if done >= train_steps:
                break
    if hip_opt:
        opt.close()
    net.eval()
    with torch.no_grad():
        tes, teo = hs.to_lists(r_te), ho.to_lists(r_te)
        net.init_history(tr, (sh, sht), (oh, oht), torch.from_numpy(te), tes, teo, te, tes, teo)
        net.latest_time = torch.from_numpy(te)[0][3]
    return net, gnet, te


def main():
    shape = sys.argv[1] if len(sys.argv) > 1 else 'ICEWS18'
    train_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    pre_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    num_k = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
    hidden = int(sys.argv[5]) if len(sys.argv) > 5 else 200
    emu = os.environ.get('RENET_BENCH_CPU_EMULATION') == '1'
    if emu:
        import cpu_abi_emulation as EMU
        EMU.install()
    dev = torch.device('cpu' if emu else 'cuda:0')
    net, gnet, te = build(shape, train_steps, pre_steps, num_k, hidden, dev, batch=64 if emu else 1024,
                           dropout=0.0 if emu else 0.2)            # (the emulation has no dropout)
    ts = np.unique(te[:, 3])
    out = {}
    for prune in (False, True):
        m = copy.deepcopy(net)
        m.prune_relations = prune
        g = torch.Generator(device='cpu').manual_seed(7)
        m.sample_entities = lambda prob, g=g, m=m: torch.multinomial(prob.detach().cpu(), m.num_k, replacement=True,
                                                                      generator=g).to(prob.device)
        secs, stats = [], []
        with torch.no_grad():
            for t in ts[1:]:
                if not emu:
                    torch.cuda.synchronize()
                t0 = time.time()
                m._advance_time(torch.tensor(int(t)), gnet)
                if not emu:
                    torch.cuda.synchronize()
                secs.append(time.time() - t0)
                stats.append(dict(m.last_prune) if m.last_prune else None)
        new_t = [t for t in m.graph_dict.keys() if t not in net.graph_dict]
        facts = {int(t): set(map(tuple, np.stack(m.graph_dict[t].global_triples(), 1).tolist())) for t in new_t}
        out[prune] = facts
        print('%-9s advance, num_k %d: %s s per timestamp%s' % (
            'pruned' if prune else 'unpruned', num_k, ['%.3f' % x for x in secs],
            ''.join('\n      object side of advance %d: %s' % (i, s) for i, s in enumerate(stats) if s)), flush=True)
    same = out[False].keys() == out[True].keys() and all(out[False][t] == out[True][t] for t in out[False])
    nf = sum(len(v) for v in out[False].values())
    diff = sum(len(out[False][t] ^ out[True].get(t, set())) for t in out[False])
    print('predicted graphs identical: %s (%d facts, symmetric difference %d)' % (same, nf, diff))


if __name__ == '__main__':
    main()
