"""CPU: the committed restatement (oracle/renet_oracle.py) against fixtures that were produced by
RUNNING THE UNMODIFIED REFERENCE (tools/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from helpers import O, fixtures, load_golden, train_case, global_shapes

RTOL, ATOL = 2e-4, 2e-5      # fp32 path, different summation order than the reference


@pytest.mark.parametrize('name', ['tiny', 'small'])
def test_history_construction_matches_reference_script(name):
    """oracle.build_histories / build_graph_dict vs data/ICEWS18/get_history_graph.py run as a script."""
    gold = load_golden('prep_%s.npz' % name)
    cfg, tr, va, te = fixtures.split_dataset(name)
    state = None
    for split, q in (('train', tr), ('valid', va), ('test', te)):
        (sh, sht), (oh, oht), state = O.build_histories(q, cfg['num_ent'], state=state)
        for tag, mine in (('s', (sh, sht)), ('o', (oh, oht))):
            ref = fixtures.unflatten_histories(gold['%s_%s_seq_ptr' % (split, tag)], gold['%s_%s_step_t' % (split, tag)],
                                               gold['%s_%s_nbr_ptr' % (split, tag)], gold['%s_%s_nbr' % (split, tag)])
            assert fixtures.histories_equal(mine, ref), (split, tag)
    gd = O.build_graph_dict(tr, cfg['num_rels'])
    assert list(gd.keys()) == gold['graph_t'].tolist()
    for k, t in enumerate(gd):
        g = gd[t]
        e0, e1 = gold['graph_edge_ptr'][k], gold['graph_edge_ptr'][k + 1]
        n0, n1 = gold['graph_node_ptr'][k], gold['graph_node_ptr'][k + 1]
        assert np.array_equal(g.ent, gold['graph_ent'][n0:n1])
        assert np.array_equal(g.src, gold['graph_src'][e0:e1])
        assert np.array_equal(g.dst, gold['graph_dst'][e0:e1])
        assert np.array_equal(g.type_s, gold['graph_type_s'][e0:e1])
        assert np.array_equal(g.type_o, gold['graph_type_o'][e0:e1])
        assert np.array_equal(g.norm(), gold['graph_norm'][n0:n1])


@pytest.mark.parametrize('d', [100, 200, 400])
def test_rgcn_layer_matches_reference(d):
    gold = load_golden('rgcn_%d.npz' % d)
    n, num_rels = int(gold['n']), int(gold['num_rels'])
    p = fixtures.make_params(200 + d, {'weight': (2 * num_rels, d * d // 100), 'loop_weight': (d, d),
                                       'h': (n, d), 'gout': (n, d)}, scale=0.5)
    for relu in (0, 1):
        for reverse in (0, 1):
            h = torch.from_numpy(p['h']).clone().requires_grad_(True)
            w = torch.from_numpy(p['weight']).clone().requires_grad_(True)
            lw = torch.from_numpy(p['loop_weight']).clone().requires_grad_(True)
            et = gold['type_o'] if reverse else gold['type_s']
            y = O.rgcn_layer(h, gold['src'], gold['dst'], et, gold['norm'], w, lw, relu=bool(relu))
            (y * torch.from_numpy(p['gout'])).sum().backward()
            tag = 'relu%d_rev%d_' % (relu, reverse)
            np.testing.assert_allclose(y.detach().numpy(), gold[tag + 'out'], rtol=RTOL, atol=ATOL)
            np.testing.assert_allclose(h.grad.numpy(), gold[tag + 'dh'], rtol=RTOL, atol=ATOL)
            for key, g in (('dweight', w.grad), ('dloop', lw.grad)):
                ok, err, how = fixtures.check_packed(gold, tag + key, g.numpy(), RTOL, ATOL * 10)
                assert ok, (tag + key, err, how)


@pytest.mark.parametrize('name,d', [('tiny', 100), ('tiny', 200), ('small', 200)])
def test_training_forward_backward_matches_reference(name, d):
    """model.RENet.forward x2 directions + backward (reference) vs renet_forward_loss (oracle)."""
    c = train_case(name, d)
    gold, cfg = c['gold'], c['cfg']
    params = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in c['params'].items()}
    gd = O.build_graph_dict(c['train'], cfg['num_rels'])
    ge = {t: torch.from_numpy(v) for t, v in c['global_emb'].items()}
    total = 0
    for tag, subject in (('s', True), ('o', False)):
        hist, hist_t = c['hists'][tag]
        loss, parts = O.renet_forward_loss(params, c['batch'], hist, hist_t, gd, ge, cfg['num_rels'],
                                           c['seq_len'], subject=subject, return_parts=True)
        assert abs(loss.item() - float(gold['loss_' + tag])) < 1e-4 * max(1.0, abs(float(gold['loss_' + tag])))
        bg = parts['bg']
        b = len(c['batch'])
        nnz = len(bg.lens)
        assert bg.num_nodes == int(gold[tag + '_graph_nodes'])
        for key, val in (('h_n', parts['s_h']), ('q_n', parts['s_q'])):
            full = np.zeros((b, d), np.float32)
            full[bg.perm] = val.detach().numpy()
            np.testing.assert_allclose(full, gold['%s_%s' % (tag, key)], rtol=RTOL, atol=ATOL)
        logits = np.zeros((b, cfg['num_ent']), np.float32)
        logits[bg.perm] = parts['ob_pred'].detach().numpy()
        np.testing.assert_allclose(logits, gold[tag + '_logits'], rtol=RTOL, atol=ATOL * 5)
        rows = parts['h2'][torch.as_tensor(bg.subj_row)].detach().numpy()
        per_seq = np.split(rows, np.cumsum(bg.lens)[:-1]) if nnz else []
        byorig = {int(bg.perm[i]): per_seq[i] for i in range(nnz)}
        mine = np.concatenate([byorig[i] for i in sorted(byorig)])
        np.testing.assert_allclose(mine, gold[tag + '_subj_rows'], rtol=RTOL, atol=ATOL)
        total = total + loss
    total.backward()
    for k, p in params.items():
        g = p.grad.numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
        ok, err, how = fixtures.check_packed(gold, 'grad.' + k, g, 1e-3, 2e-5)
        assert ok, (k, err, how)


@pytest.mark.parametrize('name,d,maxpool', [('tiny', 100, 1), ('tiny', 200, 0), ('small', 200, 1)])
def test_global_model_matches_reference(name, d, maxpool):
    gold = load_golden('global_%s_%d_max%d.npz' % (name, d, maxpool))
    cfg, tr, va, te = fixtures.split_dataset(name)
    seq_len = int(gold['seq_len'])
    p = fixtures.make_params(int(gold['param_seed']), global_shapes(cfg['num_ent'], cfg['num_rels'], d))
    params = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in p.items()}
    gd = O.build_graph_dict(tr, cfg['num_rels'])
    times = np.unique(tr[:, 3])
    loss = O.global_forward_loss(params, times, gold['true_o'], gd, seq_len, subject=True, maxpool=maxpool)
    assert abs(loss.item() - float(gold['loss'])) < 1e-4 * max(1.0, abs(float(gold['loss'])))
    loss.backward()
    for k, prm in params.items():
        if ('grad.' + k) in gold or ('grad.' + k + '__samp') in gold:
            ok, err, how = fixtures.check_packed(gold, 'grad.' + k, prm.grad.numpy(), 1e-3, 2e-5)
            assert ok, (k, err, how)
    with torch.no_grad():
        for k, t in enumerate(gold['predict_t']):
            for subj in (True, False):
                emb, logits = O.global_predict(params, int(t), gd, seq_len, subject=subj, maxpool=maxpool)
                tag = 'predict%d_%s_' % (k, 's' if subj else 'o')
                np.testing.assert_allclose(emb.numpy(), gold[tag + 'emb'], rtol=RTOL, atol=ATOL)
                np.testing.assert_allclose(logits.numpy(), gold[tag + 'logits'], rtol=RTOL, atol=ATOL)


def test_gru_restatement_matches_torch_gru():
    """gru_last_state vs torch.nn.GRU + pack_padded_sequence on CPU (the third-party op, model.py:28,86)."""
    torch.manual_seed(3)
    b, l, i, h = 9, 6, 20, 12
    gru = torch.nn.GRU(i, h, batch_first=True)
    lens = [6, 6, 5, 4, 4, 2, 1, 1, 1]
    x = torch.randn(b, l, i)
    for k, n in enumerate(lens):
        x[k, n:] = 0
    packed = torch.nn.utils.rnn.pack_padded_sequence(x, lens, batch_first=True)
    _, hn = gru(packed)
    mine = O.gru_last_state(x, lens, gru.weight_ih_l0, gru.weight_hh_l0, gru.bias_ih_l0, gru.bias_hh_l0)
    np.testing.assert_allclose(mine.detach().numpy(), hn[0].detach().numpy(), rtol=1e-5, atol=1e-6)


def test_filtered_rank_tie_rule():
    pred = np.array([0.1, 2.0, 2.0, -1.0, 2.0, 3.0], dtype=np.float32)
    # label 1 ties with 2 and 4; entity 5 is a known true completion and is filtered out
    assert O.filtered_rank(pred, 1, [5]) == 0 + (3 - 1.0) / 2 + 1
    assert O.filtered_rank(pred, 1, []) == 1 + (3 - 1.0) / 2 + 1
    m = O.mrr_hits([1, 2, 4, 20])
    assert abs(m['mrr'] - (1 + .5 + .25 + .05) / 4) < 1e-12 and m['hits@3'] == 0.5 and m['hits@10'] == 0.75


def test_oracle_batch_graph_matches_the_reference_on_degenerate_streams():
    """Direct check (only where /root/reference exists): the UNMODIFIED reference's get_big_graph +
    get_sorted_s_r_embed_rgcn (utils.py:68-93, 209-244) under the DGL shim vs oracle.batch_for_histories on tiny
    random streams with self-loop facts, exact duplicate facts and one-fact timestamps -- the cases the dataset
    fixtures rarely contain.  Compared as multisets keyed by (timestamp, entity)."""
    from collections import Counter
    from oracle import ref_loader
    if not ref_loader.available():
        pytest.skip('reference tree not present')
    ref = ref_loader.load()
    for seed in range(8):
        rng = np.random.RandomState(300 + seed)
        ne, nr, nt = int(rng.randint(3, 9)), int(rng.randint(1, 4)), int(rng.randint(3, 8))
        rows = []
        for t in range(nt):
            k = int(rng.randint(1, 7))
            q = np.stack((rng.randint(0, ne, k), rng.randint(0, nr, k), rng.randint(0, ne, k), np.full(k, t * 24)), 1)
            if rng.rand() < 0.6:
                q[0, 2] = q[0, 0]
            if k > 1 and rng.rand() < 0.6:
                q[1] = q[0]
            rows.append(q)
        tr = np.concatenate(rows).astype(np.int64)
        ogd = O.build_graph_dict(tr, nr)
        (sh, sht), _, _ = O.build_histories(tr, ne)
        with ref_loader.cpu_mode():
            rgd = {}
            for t in np.unique(tr[:, 3]):
                rgd[int(t)] = ref.utils.get_big_graph(tr[tr[:, 3] == t][:, :3], nr)
            d = 4
            ent = torch.randn(ne, d)
            glob = {int(t): torch.zeros(1, 1, d) for t in np.unique(tr[:, 3])}
            s_t, r_t = torch.from_numpy(tr[:, 0]), torch.from_numpy(tr[:, 1])
            if sum(len(h) for h in sh) == 0:
                continue
            lens_r, s_tem, r_tem, g, node_ids, _ = ref.utils.get_sorted_s_r_embed_rgcn((sh, sht), s_t, r_t, ent, rgd, glob)
            # timestamps of the member graphs in the reference's batch order (utils.py:149-170), via the same calls
            hl = torch.LongTensor(list(map(len, sh)))
            _, s_idx = hl.sort(0, descending=True)
            nz = int((hl > 0).sum())
            neighs_t = ref.utils.get_neighs_by_t([sh[i] for i in s_idx[:nz]], [sht[i] for i in s_idx[:nz]], s_t[s_idx])
            r_times = [int(t) for t in neighs_t.keys()]
        bg = O.batch_for_histories(sh, sht, tr[:, 0], ogd, sort=True)
        nnz = len(bg.lens)       # torch.sort is not stable: ties (and the empty tail) may come in another order
        assert lens_r.tolist() == bg.lens.tolist()
        assert sorted(zip(lens_r.tolist(), s_tem[:nnz].tolist())) == sorted(zip(bg.lens.tolist(), tr[bg.perm[:nnz], 0].tolist()))
        # node key = (timestamp of the member graph, entity)
        r_ent = g.ndata['id'].view(-1).tolist()
        r_graph = np.repeat(np.asarray(r_times), g.batch_num_nodes).tolist()
        o_graph = np.repeat(np.asarray(bg.graph_t), np.diff(bg.graph_off + [bg.num_nodes])).tolist()
        rk = list(zip(r_graph, r_ent))
        ok = list(zip(o_graph, bg.ent.tolist()))
        assert sorted(rk) == sorted(ok)
        rsrc, rdst = g._src.tolist(), g._dst.tolist()
        re = Counter((rk[a], rk[b], int(ts), int(to)) for a, b, ts, to in
                     zip(rsrc, rdst, g.edata['type_s'].view(-1).tolist(), g.edata['type_o'].view(-1).tolist()))
        oe = Counter((ok[a], ok[b], int(ts), int(to)) for a, b, ts, to in zip(bg.src, bg.dst, bg.type_s, bg.type_o))
        assert re == oe, (seed, re - oe, oe - re)
        rn = dict(zip(rk, g.ndata['norm'].view(-1).tolist()))
        assert all(abs(rn[k] - float(v)) < 1e-7 for k, v in zip(ok, bg.norm))
        assert Counter(rk[i] for i in node_ids) == Counter(ok[i] for i in bg.subj_row)


def test_update_cache_matches_the_reference_method():
    """RENet.update_cache (model.py:421-446) called directly on the unmodified reference class (only where
    /root/reference exists) vs ours, over random (cache, r, candidates) sequences: same rows in the same order."""
    from oracle import ref_loader
    if not ref_loader.available():
        pytest.skip('reference tree not present')
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 're-net_amd'))
    import model as M
    ref = ref_loader.load()

    class Stub(object):
        in_dim = 7
    mine_f = M.RENet.update_cache
    ref_f = ref.model.RENet.update_cache
    rng = np.random.RandomState(0)
    for trial in range(30):
        mine, theirs = [], []
        for step in range(int(rng.randint(1, 6))):
            r = int(rng.randint(0, 3))
            cand = rng.randint(0, 20, 1)          # one candidate per call (model.py:255-256); >= in_dim: the modulo
            mine = mine_f(Stub(), mine, r, cand)
            with ref_loader.cpu_mode():
                theirs = ref_f(Stub(), theirs if len(theirs) else [], torch.tensor(r), torch.from_numpy(cand))
            assert np.array_equal(np.asarray(mine), theirs.numpy()), (trial, step, mine, theirs)


# ---------------------------------------------------------------------------------------------
# config-scale: the oracle against the UNMODIFIED reference at BASELINE.json's sizes
# (tools/make_config_golden.py; ICEWS18-shaped N_ent 23 033 / R 256 / B 1024, WIKI-, GDELT-shaped, D=400 L=15)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name', ['icews18_d200', 'wiki_d200', 'gdelt_d200', 'yago_d400_l15'])
def test_oracle_matches_reference_at_config_scale(name):
    from oracle import config_cases as C
    gold = load_golden('config_%s.npz' % name)
    case = C.build_case(name, gold=gold)
    loss_s, loss_o, params, parts = C.oracle_step(case, return_parts=True)
    for tag, loss in (('s', loss_s), ('o', loss_o)):
        ref = float(gold['loss_' + tag])
        assert abs(loss.item() - ref) < 1e-5 * abs(ref), (tag, loss.item(), ref)
        assert parts[tag]['bg'].num_nodes == int(gold[tag + '_graph_nodes'])
        assert len(parts[tag]['bg'].src) == int(gold[tag + '_graph_edges'])
        un = C.unsort(parts[tag], case['spec']['batch'])
        for key in ('h_n', 'q_n', 'logits'):
            ok, err, scale = C.compare_packed(gold, '%s_%s' % (tag, key), un[key], rel=2e-4)
            assert ok, (tag, key, err, scale)
    for k, p in params.items():
        g = p.grad.numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
        ok, err, scale = C.compare_packed(gold, 'grad.' + k, g, rel=1e-3)
        assert ok, (k, err, scale)


# ---------------------------------------------------------------------------------------------
# global model at PRETRAIN scale: all 240 full graphs of the ICEWS18-shaped stream in one RGCN pass
# (N 417 704 / E 743 216), pretrain.py:82 -- oracle vs the UNMODIFIED reference's recorded outputs
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name', ['global_icews18_d200'])
def test_oracle_global_model_matches_reference_at_pretrain_scale(name):
    from oracle import config_cases as C
    gold = load_golden('config_%s.npz' % name)
    case = C.build_global_case(name)
    loss, params, ogd = C.oracle_global_step(case, subject=True)
    ref = float(gold['loss'])
    assert abs(loss.item() - ref) < 1e-5 * abs(ref), (loss.item(), ref)
    n = sum(ogd[int(t)].num_nodes for t in case['times'][:-1])          # graphs strictly before the last target time
    e = sum(len(ogd[int(t)].src) for t in case['times'][:-1])
    assert (n, e) == (int(gold['graph_nodes']), int(gold['graph_edges']))
    for k, p in params.items():
        if ('grad.' + k) in gold or ('grad.' + k + '__samp') in gold:
            ok, err, scale = C.compare_packed(gold, 'grad.' + k, p.grad.numpy(), rel=1e-3)
            assert ok, (k, err, scale)
    # get_global_emb (global_model.py:57-73) at sampled positions of the timeline
    import torch
    p0 = {k: torch.from_numpy(v) for k, v in case['params'].items()}
    keys = gold['global_emb_keys']
    times = case['times']
    unit = case['time_unit']
    rows = []
    with torch.no_grad():
        for k in (0, 1, 7, len(keys) // 2, len(keys) - 1):
            # entry keyed by times[k] is the embedding predict() returns for the NEXT timestamp
            nxt = int(times[k + 1]) if k + 1 < len(times) else int(times[-1] + unit)
            assert int(keys[k]) == int(times[k])
            s_q, _ = O.global_predict(p0, nxt, ogd, case['spec']['seq_len'], subject=True, maxpool=case['spec']['maxpool'])
            rows.append((k, s_q.numpy()))
    full = np.zeros((len(keys), case['spec']['hidden']), np.float32)
    idx = fixtures.sample_idx(full.size)
    samp = np.asarray(gold['global_emb_vals__samp'])
    for k, v in rows:
        sel = np.nonzero(idx // full.shape[1] == k)[0]
        assert len(sel)
        np.testing.assert_allclose(v[idx[sel] % full.shape[1]], samp[sel], rtol=2e-4, atol=2e-5)


def test_oracle_train_mode_applies_the_five_dropout_sites():
    """bench.py's cpu_baseline times the oracle in TRAIN mode (dropout at RGCN.py:36-37 x2, Aggregator.py:157-158,
    model.py:90,99), as the GPU step it stands beside runs; dropout = 0 is the eval-mode path the golden vectors pin."""
    import torch
    from helpers import O, train_case
    c = train_case('tiny', 200)
    cfg = c['cfg']
    params = {k: torch.from_numpy(v) for k, v in c['params'].items()}
    ogd = O.build_graph_dict(c['train'], cfg['num_rels'])
    ge = {t: torch.from_numpy(v) for t, v in c['global_emb'].items()}
    args = (params, c['batch'], c['hists']['s'][0], c['hists']['s'][1], ogd, ge, cfg['num_rels'], c['seq_len'])
    base = float(O.renet_forward_loss(*args, subject=True))
    assert float(O.renet_forward_loss(*args, subject=True, dropout=0.0)) == base
    torch.manual_seed(0)
    a = float(O.renet_forward_loss(*args, subject=True, dropout=0.5))
    torch.manual_seed(1)
    b = float(O.renet_forward_loss(*args, subject=True, dropout=0.5))
    assert a != base and a != b and all(x == x and abs(x) < 1e4 for x in (a, b))       # masks drawn, finite
