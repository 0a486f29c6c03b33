#!/usr/bin/env python
"""Generates tests/golden/*.npz by RUNNING THE UNMODIFIED REFERENCE (/root/reference) on CPU under
oracle/dgl_shim.py.  Runs only in the build container (the reference tree does not travel to the
GPU box); the fixtures it writes are committed and are what pins oracle/renet_oracle.py and the
HIP path.

    python tools/make_golden.py            # rewrites every fixture

Fixtures (inputs are regenerated from seeds by oracle/fixtures.py; only reference OUTPUTS and the
small integer inputs are stored):
  prep_<name>.npz     data/ICEWS18/get_history_graph.py run as a script on a tiny dataset:
                      histories for train/valid/test + per-timestamp graph edge lists
  rgcn_<D>.npz        RGCN.py RGCNBlockLayer forward + grads on a random multigraph
  train_<name>.npz    model.RENet.forward (both directions) + backward: losses, h_n, logits, grads
  global_<name>.npz   global_model.RENet_global forward loss + grads, predict(), get_global_emb()
"""
import os
import pickle
import runpy
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import dgl_shim, fixtures, ref_loader   # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
BIG = fixtures.BIG


def pack_tensor(out, key, t):
    """Full tensor when small; otherwise Frobenius norm + NSAMP seeded samples."""
    a = t.detach().cpu().numpy()
    if a.size <= BIG:
        out[key] = a
    else:
        out[key + '__norm'] = np.float64(np.linalg.norm(a.astype(np.float64)))
        out[key + '__samp'] = a.reshape(-1)[fixtures_sample_idx(a.size)]


def fixtures_sample_idx(n):
    return fixtures.sample_idx(n)


def run_reference_preprocessing(quads_train, quads_valid, quads_test, num_ent, num_rels):
    """Executes /root/reference/data/ICEWS18/get_history_graph.py unmodified in a temp cwd."""
    script = os.path.join(ref_loader.REFERENCE_ROOT, 'data', 'ICEWS18', 'get_history_graph.py')
    cwd = os.getcwd()
    saved = {n: sys.modules.get(n) for n in ('dgl', 'dgl.function')}
    with tempfile.TemporaryDirectory() as d:
        for name, q in (('train.txt', quads_train), ('valid.txt', quads_valid), ('test.txt', quads_test)):
            with open(os.path.join(d, name), 'w') as f:
                for s, r, o, t in q:
                    f.write('%d\t%d\t%d\t%d\t0\n' % (s, r, o, t))
        with open(os.path.join(d, 'stat.txt'), 'w') as f:
            f.write('%d\t%d\t%d\n' % (num_ent, num_rels, 0))
        dgl_shim.install()
        os.chdir(d)
        try:
            devnull = open(os.devnull, 'w')
            so = sys.stdout
            sys.stdout = devnull
            try:
                runpy.run_path(script, run_name='__main__')
            finally:
                sys.stdout = so
            res = {}
            for split, fn in (('train', 'train'), ('valid', 'dev'), ('test', 'test')):
                with open('%s_history_sub.txt' % fn, 'rb') as f:
                    sub = pickle.load(f)
                with open('%s_history_ob.txt' % fn, 'rb') as f:
                    ob = pickle.load(f)
                res[split] = (sub, ob)
            with open('train_graphs.txt', 'rb') as f:
                res['graphs'] = pickle.load(f)
        finally:
            os.chdir(cwd)
            for n, old in saved.items():
                if old is None:
                    sys.modules.pop(n, None)
                else:
                    sys.modules[n] = old
    return res


def dataset(name):
    return fixtures.split_dataset(name)


def gen_prep(name):
    cfg, tr, va, te = dataset(name)
    res = run_reference_preprocessing(tr, va, te, cfg['num_ent'], cfg['num_rels'])
    out = {}
    for split in ('train', 'valid', 'test'):
        sub, ob = res[split]
        for tag, (h, ht) in (('s', sub), ('o', ob)):
            sp, st, npz, nb = fixtures.flatten_histories(h, ht)
            out['%s_%s_seq_ptr' % (split, tag)] = sp
            out['%s_%s_step_t' % (split, tag)] = st
            out['%s_%s_nbr_ptr' % (split, tag)] = npz
            out['%s_%s_nbr' % (split, tag)] = nb
    ts, ptr, ent_ptr = [], [0], [0]
    src, dst, tys, tyo, ent, norm = [], [], [], [], [], []
    for t, g in res['graphs'].items():
        ts.append(int(t))
        src.append(g._src.numpy()); dst.append(g._dst.numpy())
        tys.append(g.edata['type_s'].numpy()); tyo.append(g.edata['type_o'].numpy())
        ent.append(g.ndata['id'].view(-1).numpy()); norm.append(g.ndata['norm'].view(-1).numpy())
        ptr.append(ptr[-1] + g.number_of_edges()); ent_ptr.append(ent_ptr[-1] + g.number_of_nodes())
    out.update(graph_t=np.asarray(ts), graph_edge_ptr=np.asarray(ptr), graph_node_ptr=np.asarray(ent_ptr),
               graph_src=np.concatenate(src), graph_dst=np.concatenate(dst),
               graph_type_s=np.concatenate(tys), graph_type_o=np.concatenate(tyo),
               graph_ent=np.concatenate(ent), graph_norm=np.concatenate(norm).astype(np.float32))
    np.savez_compressed(os.path.join(OUT, 'prep_%s.npz' % name), **out)
    return res


def gen_rgcn(d):
    """RGCNBlockLayer on a node-induced subgraph (utils.make_subgraph) of a get_big_graph graph: the
    structure RE-Net always feeds the layer -- both directions of every fact with paired types,
    multi-edges, and nodes left with zero in-degree by the induction."""
    ref = ref_loader.load()
    rng = np.random.RandomState(100 + d)
    num_ent, num_rels, m = 64, 7, 150
    trip = np.stack((rng.randint(0, num_ent, m), rng.randint(0, num_rels, m), rng.randint(0, num_ent, m)), axis=1)
    trip[:8] = trip[8:16]                              # exact duplicate facts -> multi-edges
    with ref_loader.cpu_mode():
        big = ref.utils.get_big_graph(trip, num_rels)
        present = sorted(big.ids.keys())
        keep = [e for e in present if rng.rand() < 0.75]
        g = ref.utils.make_subgraph(big, keep)
    n = g.number_of_nodes()
    out = dict(src=g._src.numpy(), dst=g._dst.numpy(), type_s=g.edata['type_s'].numpy(),
               type_o=g.edata['type_o'].numpy(), n=n, num_rels=num_rels,
               norm=g.ndata['norm'].view(-1).numpy().astype(np.float32))
    p = fixtures.make_params(200 + d, {'weight': (2 * num_rels, d * d // 100), 'loop_weight': (d, d),
                                       'h': (n, d), 'gout': (n, d)}, scale=0.5)
    for relu in (0, 1):
        for reverse in (0, 1):
            layer = ref.RGCN.RGCNBlockLayer(d, d, 2 * num_rels, 100, activation=(torch.relu if relu else None),
                                            self_loop=True, dropout=0.0)
            with torch.no_grad():
                layer.weight.copy_(torch.from_numpy(p['weight']))
                layer.loop_weight.copy_(torch.from_numpy(p['loop_weight']))
            h = torch.from_numpy(p['h']).clone().requires_grad_(True)
            g.ndata['h'] = h
            layer(g, bool(reverse))
            y = g.ndata.pop('h')
            (y * torch.from_numpy(p['gout'])).sum().backward()
            tag = 'relu%d_rev%d_' % (relu, reverse)
            out[tag + 'out'] = y.detach().numpy()
            out[tag + 'dh'] = h.grad.numpy()
            pack_tensor(out, tag + 'dweight', layer.weight.grad)
            pack_tensor(out, tag + 'dloop', layer.loop_weight.grad)
    np.savez_compressed(os.path.join(OUT, 'rgcn_%d.npz' % d), **out)


def ref_graph_dict_from(res):
    return res['graphs']


def gen_train(name, d, seq_len, batch_size, prep):
    ref = ref_loader.load()
    cfg, tr, va, te = dataset(name)
    num_ent, num_rels = cfg['num_ent'], cfg['num_rels']
    (s_hist, s_hist_t), (o_hist, o_hist_t) = prep['train']
    graph_dict = prep['graphs']
    model = ref.model.RENet(num_ent, d, num_rels, dropout=0.0, model=0, seq_len=seq_len, num_k=10)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    params = fixtures.make_params(cfg['seed'] * 7 + d, shapes, scale=None)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    times = np.unique(tr[:, 3])
    gl = fixtures.make_params(cfg['seed'] * 11 + d, {'g': (len(times), d)}, scale=0.3)['g']
    model.global_emb = {int(t): torch.from_numpy(gl[k]).view(1, 1, d) for k, t in enumerate(times)}
    # batch: a seeded mix of early (empty-history) and late triples; histories truncated to seq_len
    # exactly as a reference run with history_len == seq_len would produce them
    rng = np.random.RandomState(cfg['seed'] + 5)
    idx = np.sort(rng.choice(len(tr), size=batch_size, replace=False))
    idx[:3] = [0, 1, 2]                                    # guaranteed empty histories
    cut = lambda h: [x[-seq_len:] for x in h]
    bs_h, bs_t = cut([s_hist[i] for i in idx]), cut([s_hist_t[i] for i in idx])
    bo_h, bo_t = cut([o_hist[i] for i in idx]), cut([o_hist_t[i] for i in idx])
    batch = torch.from_numpy(tr[idx]).long()
    out = dict(batch_idx=idx, d=d, seq_len=seq_len, param_seed=cfg['seed'] * 7 + d,
               global_seed=cfg['seed'] * 11 + d)
    cap = {}
    hooks = [model.encoder.register_forward_hook(lambda m, i, o: cap.setdefault('enc', []).append(o[1].detach().clone())),
             model.encoder_r.register_forward_hook(lambda m, i, o: cap.setdefault('enc_r', []).append(o[1].detach().clone())),
             model.linear.register_forward_hook(lambda m, i, o: cap.setdefault('lin', []).append(o.detach().clone())),
             model.aggregator.rgcn2.register_forward_hook(
                 lambda m, i, o: cap.setdefault('h2', []).append((o.ndata['h'].detach().clone(),
                                                                  o.ndata['id'].view(-1).clone(),
                                                                  list(o.batch_num_nodes))))]
    orig = ref.Aggregator.get_sorted_s_r_embed_rgcn

    def spy(*a, **k):
        r = orig(*a, **k)
        cap.setdefault('rows', []).append(list(r[4]))
        cap.setdefault('lens', []).append(r[0].clone())
        return r
    ref.Aggregator.get_sorted_s_r_embed_rgcn = spy
    try:
        with ref_loader.cpu_mode():
            loss_s = model(batch, (bs_h, bs_t), (bo_h, bo_t), graph_dict, subject=True)
            loss_o = model(batch, (bs_h, bs_t), (bo_h, bo_t), graph_dict, subject=False)
            (loss_s + loss_o).backward()
    finally:
        ref.Aggregator.get_sorted_s_r_embed_rgcn = orig
        for h in hooks:
            h.remove()
    out['loss_s'] = np.float64(loss_s.item())
    out['loss_o'] = np.float64(loss_o.item())
    for k, p in model.named_parameters():
        pack_tensor(out, 'grad.' + k, p.grad)
    for di, (tag, hist) in enumerate((('s', bs_h), ('o', bo_h))):
        lens = torch.LongTensor([len(h) for h in hist])
        _, perm = lens.sort(0, descending=True)            # the permutation model.py:81 computed
        perm = perm.numpy()
        nnz = int((lens > 0).sum())
        for key, capk in (('h_n', 'enc'), ('q_n', 'enc_r')):
            hn = cap[capk][di].view(-1, d).numpy()
            full = np.zeros((batch_size, d), np.float32)
            full[perm[:nnz]] = hn
            out['%s_%s' % (tag, key)] = full
        logits = cap['lin'][di].numpy()
        un = np.zeros_like(logits)
        un[perm] = logits
        out['%s_logits' % tag] = un
        # subject rows of h2, re-keyed by ORIGINAL batch position then step
        h2, ids, counts = cap['h2'][di]
        rows = h2[torch.LongTensor(cap['rows'][di])].numpy()
        lens_sorted = cap['lens'][di].numpy()
        per_seq = np.split(rows, np.cumsum(lens_sorted)[:-1]) if nnz else []
        byorig = {int(perm[i]): per_seq[i] for i in range(nnz)}
        out['%s_subj_rows' % tag] = (np.concatenate([byorig[i] for i in sorted(byorig)])
                                     if nnz else np.zeros((0, d), np.float32))
        out['%s_graph_nodes' % tag] = np.int64(h2.shape[0])
    np.savez_compressed(os.path.join(OUT, 'train_%s_%d.npz' % (name, d)), **out)


def gen_global(name, d, seq_len, prep, maxpool):
    ref = ref_loader.load()
    cfg, tr, va, te = dataset(name)
    num_ent, num_rels = cfg['num_ent'], cfg['num_rels']
    graph_dict = prep['graphs']
    model = ref.global_model.RENet_global(num_ent, d, num_rels, dropout=0.0, model=0, seq_len=seq_len,
                                          num_k=10, maxpool=maxpool)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    params = fixtures.make_params(cfg['seed'] * 13 + d, shapes, scale=None)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    times = np.unique(tr[:, 3])
    true_s, true_o = ref.utils.get_true_distribution(tr, num_ent)
    out = dict(d=d, seq_len=seq_len, maxpool=maxpool, param_seed=cfg['seed'] * 13 + d,
               true_s=true_s, true_o=true_o)
    t_list = torch.from_numpy(times.copy())
    with ref_loader.cpu_mode():
        loss = model(t_list, torch.from_numpy(true_s), torch.from_numpy(true_o), graph_dict, subject=True)
        loss.backward()
        out['loss'] = np.float64(loss.item())
        for k, p in model.named_parameters():
            if p.grad is not None:
                pack_tensor(out, 'grad.' + k, p.grad)
        with torch.no_grad():
            pts = [int(times[1]), int(times[len(times) // 2]), int(times[-1] + cfg['time_unit'])]
            out['predict_t'] = np.asarray(pts)
            for k, t in enumerate(pts):
                for subj in (True, False):
                    s_q, sub, prob = model.predict(t, graph_dict, subject=subj)
                    tag = 'predict%d_%s_' % (k, 's' if subj else 'o')
                    out[tag + 'emb'] = s_q.view(-1).numpy()
                    out[tag + 'logits'] = sub.view(-1).numpy()
            ge = model.get_global_emb(times, graph_dict)
            out['global_emb_keys'] = np.asarray(list(ge.keys()), dtype=np.int64)
            out['global_emb_vals'] = np.stack([ge[k].view(-1).numpy() for k in ge.keys()])
    np.savez_compressed(os.path.join(OUT, 'global_%s_%d_max%d.npz' % (name, d, maxpool)), **out)


def gen_eval(name, d, seq_len, num_k, n_eval, prep):
    """model.RENet.evaluate_filter over the first n_eval validation quadruples (the multi-step inference
    state machine, model.py:216-419), with the random entity samples recorded so that another
    implementation can be driven through the identical trajectory."""
    ref = ref_loader.load()
    cfg, tr, va, te = dataset(name)
    num_ent, num_rels = cfg['num_ent'], cfg['num_rels']
    import copy
    graph_dict = copy.deepcopy(prep['graphs'])
    model = ref.model.RENet(num_ent, d, num_rels, dropout=0.0, model=0, seq_len=seq_len, num_k=num_k)
    gmodel = ref.global_model.RENet_global(num_ent, d, num_rels, dropout=0.0, model=0, seq_len=seq_len,
                                           num_k=num_k, maxpool=1)
    pm = fixtures.make_params(cfg['seed'] * 17 + d, {k: tuple(v.shape) for k, v in model.state_dict().items()})
    pg = fixtures.make_params(cfg['seed'] * 19 + d, {k: tuple(v.shape) for k, v in gmodel.state_dict().items()})
    model.load_state_dict({k: torch.from_numpy(v) for k, v in pm.items()})
    gmodel.load_state_dict({k: torch.from_numpy(v) for k, v in pg.items()})
    model.eval(); gmodel.eval()
    samples = []
    Cat = torch.distributions.categorical.Categorical
    orig_sample = Cat.sample

    def rec_sample(self, shape=torch.Size()):
        out = orig_sample(self, shape)
        samples.append(out.clone())
        return out
    total = torch.from_numpy(np.concatenate((tr, va, te)))
    valid = torch.from_numpy(va)
    (vs, vst), (vo, vot) = prep['valid']
    (ts_, tst), (to_, tot) = prep['test']
    (s_hist, s_hist_t), (o_hist, o_hist_t) = prep['train']
    ranks, losses = [], []
    Cat.sample = rec_sample
    # model.py:229-297 re-uses the names `s` / `o` as loop variables, so the FIRST quadruple of every new timestamp
    # is scored for the entity of the LAST candidate `torch.topk(prob_tensor, num_k, sorted=False)` returned
    # (s = s_to_id[indices[-1]], model.py:254-255 / 291-292) instead of its own subject / object.  The order of an
    # unsorted top-k is device dependent, so the shadowing entities are recorded per advanced timestamp.
    cand_topk = []
    orig_topk = torch.topk

    def rec_topk(inp, k, *a, **kw):
        out = orig_topk(inp, k, *a, **kw)
        if inp.dim() == 1 and inp.numel() == num_k * num_k and k == num_k:
            cand_topk.append(out[1].clone())
        return out
    torch.topk = rec_topk
    try:
        with ref_loader.cpu_mode(), torch.no_grad():
            times = np.unique(tr[:, 3])
            model.global_emb = gmodel.get_global_emb(times, graph_dict)
            model.graph_dict = graph_dict
            model.init_history(tr, (s_hist, s_hist_t), (o_hist, o_hist_t), valid, (vs, vst), (vo, vot), te,
                               (ts_, tst), (to_, tot))
            model.latest_time = valid[0][3]
            for i in range(n_eval):
                rk, loss = model.evaluate_filter(valid[i], (vs[i], vst[i]), (vo[i], vot[i]), gmodel, total)
                ranks.append(rk)
                losses.append(loss.item())
    finally:
        Cat.sample = orig_sample
        torch.topk = orig_topk
    # per advanced timestamp: samples[2a] = subjects, samples[2a+1] = objects; cand_topk likewise
    assert len(cand_topk) == len(samples) and len(samples) % 2 == 0
    shadow = np.asarray([[int(samples[2 * a + side][int(cand_topk[2 * a + side][-1]) // num_k]) for side in (0, 1)]
                         for a in range(len(samples) // 2)], dtype=np.int64).reshape(-1, 2)
    out = dict(d=d, seq_len=seq_len, num_k=num_k, n_eval=n_eval, model_seed=cfg['seed'] * 17 + d, shadow=shadow,
               global_seed=cfg['seed'] * 19 + d, ranks=np.asarray(ranks), losses=np.asarray(losses),
               samples=np.stack([x.numpy() for x in samples]) if samples else np.zeros((0, num_k), np.int64),
               n_new_graphs=np.int64(len(graph_dict) - len(prep['graphs'])))
    # the graphs the model predicted for the timestamps it advanced over (model.py:300-301)
    new_t = [t for t in graph_dict if t not in prep['graphs']]
    trip = []
    for t in new_t:
        g = graph_dict[t]
        m = g.number_of_edges() // 2
        ids = g.ndata['id'].view(-1).numpy()
        q = np.stack((ids[g._src[:m].numpy()], g.edata['type_s'][:m].numpy(), ids[g._dst[:m].numpy()],
                      np.full(m, int(t))), axis=1)
        trip.append(q[np.lexsort((q[:, 2], q[:, 1], q[:, 0]))])
    out['new_graph_quads'] = np.concatenate(trip) if trip else np.zeros((0, 4), np.int64)
    np.savez_compressed(os.path.join(OUT, 'eval_%s_%d.npz' % (name, d)), **out)


def main():
    if not ref_loader.available():
        raise SystemExit('reference tree not available: fixtures can only be generated in the build container')
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    for d in (100, 200, 400):
        gen_rgcn(d)
    preps = {name: gen_prep(name) for name in fixtures.DATASETS}
    gen_train('tiny', 100, 4, 40, preps['tiny'])
    gen_train('tiny', 200, 10, 40, preps['tiny'])
    gen_train('small', 200, 10, 96, preps['small'])
    gen_global('tiny', 100, 4, preps['tiny'], 1)
    gen_global('tiny', 200, 10, preps['tiny'], 0)
    gen_global('small', 200, 10, preps['small'], 1)
    gen_eval('small', 100, 10, 6, 180, preps['small'])
    for f in sorted(os.listdir(OUT)):
        print('%-32s %8.1f KB' % (f, os.path.getsize(os.path.join(OUT, f)) / 1024.0))


if __name__ == '__main__':
    main()
