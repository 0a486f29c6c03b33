#!/usr/bin/env python
"""Config-scale golden fixtures: ONE eval-mode training step (model.RENet.forward for both directions +
backward, train.py:136-138) of the UNMODIFIED reference (/root/reference under oracle/dgl_shim.py, CPU) on
the cases of oracle/config_cases.py -- the bench workload (ICEWS18-shaped, N_ent 23 033, R 256, B 1024), the
WIKI- and GDELT-shaped streams and YAGO-shaped n_hidden=400 / seq_len=15.  Build container only.

    python tools/make_config_golden.py [case ...]

Writes tests/golden/config_<case>.npz: losses, graph sizes, and for every big tensor (h_n, q_n, logits in
original batch order, the gradient of every parameter) its Frobenius norm + 4096 seeded samples
(tools/make_golden.py:pack_tensor), plus the batch's flattened histories -- a few hundred KB per case.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))

from oracle import config_cases as C, fixtures, ref_loader   # noqa: E402
from make_golden import pack_tensor                # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')


def gen(name):
    ref = ref_loader.load()
    t0 = time.time()
    case = C.build_case(name)
    spec, quads, num_ent, num_rels = case['spec'], case['quads'], case['num_ent'], case['num_rels']
    d, seq_len, B = spec['hidden'], spec['seq_len'], spec['batch']
    print('%s: %d quads, histories built in %.0f s' % (name, len(quads), time.time() - t0), flush=True)
    out = dict(idx=case['idx'], d=d, seq_len=seq_len)
    for tag in ('s', 'o'):       # the batch's histories (oracle.build_histories, pinned by tests/golden/prep_*.npz)
        sp, st, npt, nb = fixtures.flatten_histories(*case['hists'][tag])
        out.update({'hist_%s_seq_ptr' % tag: sp, 'hist_%s_step_t' % tag: st, 'hist_%s_nbr_ptr' % tag: npt,
                    'hist_%s_nbr' % tag: nb})
    cap = {}
    with ref_loader.cpu_mode():
        graph_dict = {}
        order = np.argsort(quads[:, 3], kind='stable')
        q = quads[order]
        times, starts = np.unique(q[:, 3], return_index=True)
        ends = np.concatenate((starts[1:], [len(q)]))
        for t, a, b in zip(times, starts, ends):                       # data/*/get_history_graph.py:137-140
            graph_dict[int(t)] = ref.utils.get_big_graph(q[a:b, :3], num_rels)
        model = ref.model.RENet(num_ent, d, num_rels, dropout=0.0, model=0, seq_len=seq_len, num_k=10)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in case['params'].items()})
        model.global_emb = {t: torch.from_numpy(v).view(1, 1, d) for t, v in case['global_emb'].items()}
        model.eval()
        hooks = [model.encoder.register_forward_hook(lambda m, i, o: cap.setdefault('enc', []).append(o[1].detach().clone())),
                 model.encoder_r.register_forward_hook(lambda m, i, o: cap.setdefault('enc_r', []).append(o[1].detach().clone())),
                 model.linear.register_forward_hook(lambda m, i, o: cap.setdefault('lin', []).append(o.detach().clone())),
                 model.aggregator.rgcn2.register_forward_hook(
                     lambda m, i, o: cap.setdefault('gsz', []).append((int(o.number_of_nodes()), int(o.number_of_edges()))))]
        batch = torch.from_numpy(case['batch']).long()
        hs, ho = case['hists']['s'], case['hists']['o']
        t0 = time.time()
        loss_s = model(batch, hs, ho, graph_dict, subject=True)
        loss_o = model(batch, hs, ho, graph_dict, subject=False)
        (loss_s + loss_o).backward()
        print('  reference step: %.1f s, loss_s %.6f loss_o %.6f' % (time.time() - t0, loss_s.item(), loss_o.item()),
              flush=True)
        for h in hooks:
            h.remove()
    out['loss_s'] = np.float64(loss_s.item())
    out['loss_o'] = np.float64(loss_o.item())
    for k, p in model.named_parameters():
        pack_tensor(out, 'grad.' + k, p.grad)
    for di, (tag, hist) in enumerate((('s', hs[0]), ('o', ho[0]))):
        lens = torch.LongTensor([len(h) for h in hist])
        _, perm = lens.sort(0, descending=True)                        # the permutation model.py:81 computed
        perm = perm.numpy()
        nnz = int((lens > 0).sum())
        for key, capk in (('h_n', 'enc'), ('q_n', 'enc_r')):
            hn = cap[capk][di].view(-1, d)
            full = torch.zeros(B, d)
            full[torch.from_numpy(perm[:nnz])] = hn
            pack_tensor(out, '%s_%s' % (tag, key), full)
        logits = cap['lin'][di]
        un = torch.zeros_like(logits)
        un[torch.from_numpy(perm)] = logits
        pack_tensor(out, '%s_logits' % tag, un)
        out['%s_graph_nodes' % tag] = np.int64(cap['gsz'][di][0])
        out['%s_graph_edges' % tag] = np.int64(cap['gsz'][di][1])
        out['%s_nnz' % tag] = np.int64(nnz)
        print('  %s: graph N=%d E=%d, non-empty %d' % (tag, cap['gsz'][di][0], cap['gsz'][di][1], nnz), flush=True)
    np.savez_compressed(os.path.join(OUT, 'config_%s.npz' % name), **out)


def gen_global(name):
    """RENet_global.forward (subject pass, pretrain.py:82) + backward at pretrain scale -- every training timestamp
    in ONE batch => all full graphs in one RGCN pass -- and get_global_emb over the whole timeline
    (global_model.py:57-73), by the UNMODIFIED reference.  Writes tests/golden/config_<name>.npz."""
    ref = ref_loader.load()
    case = C.build_global_case(name)
    spec, quads, num_ent, num_rels = case['spec'], case['quads'], case['num_ent'], case['num_rels']
    d, seq_len, times = spec['hidden'], spec['seq_len'], case['times']
    out = dict(d=d, seq_len=seq_len, maxpool=spec['maxpool'], num_t=np.int64(len(times)))
    cap = {}
    with ref_loader.cpu_mode():
        graph_dict = {}
        order = np.argsort(quads[:, 3], kind='stable')
        q = quads[order]
        ts, starts = np.unique(q[:, 3], return_index=True)
        ends = np.concatenate((starts[1:], [len(q)]))
        for t, a, b in zip(ts, starts, ends):
            graph_dict[int(t)] = ref.utils.get_big_graph(q[a:b, :3], num_rels)
        model = ref.global_model.RENet_global(num_ent, d, num_rels, dropout=0.0, model=0, seq_len=seq_len, num_k=10,
                                              maxpool=spec['maxpool'])
        model.load_state_dict({k: torch.from_numpy(v) for k, v in case['params'].items()})
        model.eval()
        hook = model.aggregator.rgcn2.register_forward_hook(
            lambda m, i, o: cap.setdefault('gsz', []).append((int(o.number_of_nodes()), int(o.number_of_edges()))))
        enc = model.encoder_global.register_forward_hook(lambda m, i, o: cap.setdefault('enc', []).append(o[1].detach().clone()))
        t0 = time.time()
        loss = model(torch.from_numpy(times.copy()), torch.from_numpy(case['true_s']), torch.from_numpy(case['true_o']),
                     graph_dict, subject=True)
        loss.backward()
        hook.remove()
        enc.remove()
        print('%s: reference pretrain step %.1f s, loss %.6f, graph N=%d E=%d'
              % (name, time.time() - t0, loss.item(), cap['gsz'][0][0], cap['gsz'][0][1]), flush=True)
        out['loss'] = np.float64(loss.item())
        out['graph_nodes'], out['graph_edges'] = np.int64(cap['gsz'][0][0]), np.int64(cap['gsz'][0][1])
        pack_tensor(out, 's_q_sorted', cap['enc'][0].view(-1, d))      # rows: timestamps descending, t = 0 dropped
        for k, p in model.named_parameters():
            if p.grad is not None:
                pack_tensor(out, 'grad.' + k, p.grad)
        with torch.no_grad():
            t0 = time.time()
            ge = model.get_global_emb(times, graph_dict)
            out['global_emb_keys'] = np.asarray([int(k) for k in ge.keys()], dtype=np.int64)
            pack_tensor(out, 'global_emb_vals', torch.stack([ge[k].view(-1) for k in ge.keys()]))
            print('  get_global_emb: %.1f s, %d entries' % (time.time() - t0, len(ge)), flush=True)
    np.savez_compressed(os.path.join(OUT, 'config_%s.npz' % name), **out)


def gen_eval(name):
    """test.py's evaluation loop (model.RENet.evaluate_filter, model.py:216-419) of the UNMODIFIED reference at
    config scale: N_ent 23 033, R 256, num_k 1000, over quadruples of the first validation timestamps -- every
    timestamp advance samples 2 x 1000 entities and scores a [256 x 23 033] joint distribution for each.  The
    random entity samples, the shadowing entities (unsorted-top-k order, see tools/make_golden.py:gen_eval) and the
    predicted graphs are recorded.  Writes tests/golden/config_<name>.npz."""
    import copy
    from oracle import renet_oracle as O
    ref = ref_loader.load()
    case = C.build_eval_case(name)
    spec, num_ent, num_rels = case['spec'], case['num_ent'], case['num_rels']
    d, seq_len, num_k = spec['hidden'], spec['seq_len'], spec['num_k']
    tr, va, te = case['train'], case['valid'], case['test']
    t0 = time.time()
    (s_hist, s_hist_t), (o_hist, o_hist_t), st = O.build_histories(tr, num_ent, history_len=seq_len)
    (vs, vst), (vo, vot), st = O.build_histories(va, num_ent, history_len=seq_len, state=st)
    (ts_, tst), (to_, tot), st = O.build_histories(te, num_ent, history_len=seq_len, state=st)
    print('%s: histories %.0f s' % (name, time.time() - t0), flush=True)
    samples, cand_topk = [], []
    Cat = torch.distributions.categorical.Categorical
    orig_sample, orig_topk = Cat.sample, torch.topk

    def rec_sample(self, shape=torch.Size()):
        o = orig_sample(self, shape)
        samples.append(o.clone())
        print('  sampled %d entities (%d distinct) at %.0f s' % (o.numel(), len(set(o.tolist())), time.time() - t0),
              flush=True)
        return o

    def rec_topk(inp, k, *a, **kw):
        o = orig_topk(inp, k, *a, **kw)
        if inp.dim() == 1 and inp.numel() % num_k == 0 and inp.numel() <= num_k * num_k and k == num_k \
                and inp.numel() != num_rels * num_ent:
            cand_topk.append(o[1].clone())
        return o
    ranks, losses = [], []
    with ref_loader.cpu_mode(), torch.no_grad():
        graph_dict = {}
        ts, starts = np.unique(tr[:, 3], return_index=True)
        ends = np.concatenate((starts[1:], [len(tr)]))
        for t, a, b in zip(ts, starts, ends):
            graph_dict[int(t)] = ref.utils.get_big_graph(tr[a:b, :3], num_rels)
        n_graphs0 = len(graph_dict)
        model = ref.model.RENet(num_ent, d, num_rels, dropout=0.0, model=0, seq_len=seq_len, num_k=num_k)
        gmodel = ref.global_model.RENet_global(num_ent, d, num_rels, dropout=0.0, model=0, seq_len=seq_len,
                                               num_k=num_k, maxpool=1)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in case['params'].items()})
        gmodel.load_state_dict({k: torch.from_numpy(v) for k, v in case['gparams'].items()})
        model.eval(); gmodel.eval()
        total = torch.from_numpy(np.concatenate((tr, va, te)))
        valid = torch.from_numpy(va)
        model.global_emb = gmodel.get_global_emb(np.unique(tr[:, 3]), graph_dict)
        model.graph_dict = graph_dict
        model.init_history(tr, (s_hist, s_hist_t), (o_hist, o_hist_t), valid, (vs, vst), (vo, vot), te,
                           (ts_, tst), (to_, tot))
        model.latest_time = valid[0][3]
        print('  state ready at %.0f s' % (time.time() - t0), flush=True)
        Cat.sample, torch.topk = rec_sample, rec_topk
        try:
            for i in case['eval_idx']:
                rk, loss = model.evaluate_filter(valid[i], (vs[i], vst[i]), (vo[i], vot[i]), gmodel, total)
                ranks.append(rk)
                losses.append(loss.item())
                print('  quad %d: ranks %s loss %.5f at %.0f s' % (i, rk.tolist(), loss.item(), time.time() - t0),
                      flush=True)
        finally:
            Cat.sample, torch.topk = orig_sample, orig_topk
    assert len(cand_topk) == len(samples) and len(samples) % 2 == 0, (len(cand_topk), len(samples))
    # the candidate index refers to the de-duplicated key order of preds_list (dict insertion order == first
    # occurrence order of the samples): map through it
    shadow = []
    for a in range(len(samples) // 2):
        row = []
        for side in (0, 1):
            smp = samples[2 * a + side].tolist()
            keys = list(dict.fromkeys(smp))
            # model.py:229-235 keys preds_list by TENSOR objects (identity hash): every sample is its own key,
            # duplicates included -- the candidate index therefore addresses the raw sample list
            row.append(int(smp[int(cand_topk[2 * a + side][-1]) // num_k]))
        shadow.append(row)
    out = dict(d=d, seq_len=seq_len, num_k=num_k, eval_idx=case['eval_idx'], shadow=np.asarray(shadow, np.int64).reshape(-1, 2),
               ranks=np.asarray(ranks), losses=np.asarray(losses),
               samples=np.stack([x.numpy() for x in samples]), n_new_graphs=np.int64(len(graph_dict) - n_graphs0))
    trip = []
    for t in list(graph_dict.keys())[n_graphs0:]:
        g = graph_dict[t]
        m = g.number_of_edges() // 2
        ids = g.ndata['id'].view(-1).numpy()
        q = np.stack((ids[g._src[:m].numpy()], g.edata['type_s'][:m].numpy(), ids[g._dst[:m].numpy()],
                      np.full(m, int(t))), axis=1)
        trip.append(q[np.lexsort((q[:, 2], q[:, 1], q[:, 0]))])
    out['new_graph_quads'] = np.concatenate(trip) if trip else np.zeros((0, 4), np.int64)
    np.savez_compressed(os.path.join(OUT, 'config_%s.npz' % name), **out)
    print('  done: %d new graphs, %d predicted quads, %.0f s' % (out['n_new_graphs'], len(out['new_graph_quads']),
                                                                   time.time() - t0), flush=True)


if __name__ == '__main__':
    for n in (sys.argv[1:] or sorted(C.CASES) + sorted(C.GLOBAL_CASES) + sorted(C.EVAL_CASES)):
        (gen_global if n in C.GLOBAL_CASES else gen_eval if n in C.EVAL_CASES else gen)(n)
