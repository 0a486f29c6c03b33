"""The inference advance with relation pruning (RENet.prune_relations, model._winners_pruned) against the unpruned path, on
CPU with the device wrappers emulated in torch (tests/cpu_abi_emulation.py): the SAME winners must come out -- the pruning
argument is exact (an upper bound per (entity, relation) row against a lower bound of the final threshold)."""
import copy
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from helpers import fixtures, renet_shapes, global_shapes      # noqa: E402
import cpu_abi_emulation as EMU                                 # noqa: E402


def _setup(num_k, sharpen):
    import global_model as GM
    import model as M
    import preprocess as P
    import utils as U
    cfg, tr, va, te = fixtures.split_dataset('small')
    d, seq_len = 100, 5
    net = M.RENet(cfg['num_ent'], d, cfg['num_rels'], dropout=0.0, seq_len=seq_len, num_k=num_k)
    gnet = GM.RENet_global(cfg['num_ent'], d, cfg['num_rels'], dropout=0.0, seq_len=seq_len, num_k=num_k, maxpool=1)
    params = fixtures.make_params(11, renet_shapes(cfg['num_ent'], cfg['num_rels'], d))
    for k in ('linear.weight', 'linear_r.weight'):
        params[k] = params[k] * sharpen                       # peaked p(o | s, r) and p(r | s): pruning has something to cut
    net.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    gnet.load_state_dict({k: torch.from_numpy(v) for k, v in
                          fixtures.make_params(12, global_shapes(cfg['num_ent'], cfg['num_rels'], d)).items()})
    net.eval()
    gnet.eval()
    allq = np.concatenate((tr, va, te))
    hs, ho = P.HistoryIndex(allq, 's', seq_len), P.HistoryIndex(allq, 'o', seq_len)
    rng = {'train': np.arange(0, len(tr)), 'valid': np.arange(len(tr), len(tr) + len(va)),
           'test': np.arange(len(tr) + len(va), len(allq))}
    H = {k: (hs.to_lists(v), ho.to_lists(v)) for k, v in rng.items()}
    gd = U.build_graph_dict(tr, cfg['num_rels'])
    valid = torch.from_numpy(va)
    with torch.no_grad():
        net.global_emb = gnet.get_global_emb(np.unique(tr[:, 3]), gd)
        net.graph_dict = gd
        net.init_history(tr, H['train'][0], H['train'][1], valid, H['valid'][0], H['valid'][1], te,
                         H['test'][0], H['test'][1])
        net.latest_time = valid[0][3]
    return net, cfg


def _cache_sets(cache):
    return [set(map(tuple, np.asarray(c, dtype=np.int64).reshape(-1, 2).tolist())) for c in cache]


@pytest.mark.parametrize('sharpen,block_log2', [(40.0, 10), (40.0, 28), (1.0, 11), (8.0, 12)])
def test_pruned_advance_selects_the_same_winners(sharpen, block_log2, monkeypatch):
    undo = EMU.install()
    try:
        monkeypatch.setenv('RENET_ADVANCE_BLOCK_LOG2', str(block_log2))
        num_k = 24
        net, cfg = _setup(num_k, sharpen)
        n_ent, R = cfg['num_ent'], cfg['num_rels']
        rs = np.random.RandomState(3)
        for subject in (True, False):
            # a peaked sampling distribution with REPEATED picks (the reference counts an entity once per pick)
            logits = torch.from_numpy(rs.randn(n_ent).astype(np.float32) * 3.0)
            prob = torch.softmax(logits, dim=0)
            picks = torch.multinomial(prob, num_k, replacement=True)
            assert len(np.unique(picks.numpy())) < num_k
            a, b = copy.deepcopy(net), copy.deepcopy(net)
            a.prune_relations, b.prune_relations = False, True
            with torch.no_grad():
                a._advance_side(picks, prob, subject)
                b._advance_side(picks, prob, subject)
            ca, cb = (a.s_his_cache, b.s_his_cache) if subject else (a.o_his_cache, b.o_his_cache)
            sa, sb = _cache_sets(ca), _cache_sets(cb)
            n_facts = sum(len(x) for x in sa)
            assert n_facts > 0
            diff = sum(len(x ^ y) for x, y in zip(sa, sb))
            # equal values at the threshold may be resolved differently (torch.topk on different candidate lists): none seen
            assert diff == 0, (subject, sharpen, n_facts, diff)
            ta, tb = (a.s_his_cache_t, b.s_his_cache_t) if subject else (a.o_his_cache_t, b.o_his_cache_t)
            assert list(ta) == list(tb)
            st = b.last_prune
            assert st['rows'] == len(np.unique(picks.numpy())) * R and 0 < st['scored'] <= st['rows']
            if sharpen >= 40.0 and block_log2 == 10:
                assert st['scored'] < st['rows'], st          # peaked distributions, small blocks: rows WERE skipped
    finally:
        undo()


def test_pruned_advance_through_the_evaluation_loop(monkeypatch):
    """evaluate_filter over a few validation quadruples that cross a timestamp boundary (one full _advance_time: both sides,
    predicted graph, rolled histories) with and without pruning: same predicted graph, same ranks."""
    undo = EMU.install()
    try:
        monkeypatch.setenv('RENET_ADVANCE_BLOCK_LOG2', '11')
        import global_model as GM
        num_k = 16
        net, cfg = _setup(num_k, 20.0)
        gnet = GM.RENet_global(cfg['num_ent'], 100, cfg['num_rels'], dropout=0.0, seq_len=5, num_k=num_k, maxpool=1)
        gnet.load_state_dict({k: torch.from_numpy(v) for k, v in
                              fixtures.make_params(12, global_shapes(cfg['num_ent'], cfg['num_rels'], 100)).items()})
        gnet.eval()
        _, tr, va, te = fixtures.split_dataset('small')
        import preprocess as P
        allq = np.concatenate((tr, va, te))
        hs, ho = P.HistoryIndex(allq, 's', 5), P.HistoryIndex(allq, 'o', 5)
        idx = np.arange(len(tr), len(tr) + len(va))
        (vs, vst), (vo, vot) = hs.to_lists(idx), ho.to_lists(idx)
        total = torch.from_numpy(allq)
        valid = torch.from_numpy(va)
        t_first = int(va[0, 3])
        n_eval = int(np.count_nonzero(va[:, 3] == t_first)) + 3          # into the second timestamp: one advance
        out = []
        for prune in (False, True):
            m = copy.deepcopy(net)
            m.prune_relations = prune
            torch.manual_seed(5)                                           # sample_entities draws from torch's generator
            with torch.no_grad():
                ranks = [m.evaluate_filter(valid[i], (vs[i], vst[i]), (vo[i], vot[i]), gnet, total)[0] for i in range(n_eval)]
            new_t = [t for t in m.graph_dict.keys() if t not in net.graph_dict]
            facts = {t: set(map(tuple, np.stack(m.graph_dict[t].global_triples(), 1).tolist())) for t in new_t}
            out.append((np.asarray(ranks), facts, m.last_prune))
        assert out[1][2] is not None and out[0][2] is None
        assert out[0][1].keys() == out[1][1].keys() and len(out[0][1]) >= 1
        for t in out[0][1]:
            assert out[0][1][t] == out[1][1][t]
        assert np.array_equal(out[0][0], out[1][0])
    finally:
        undo()


def test_lookahead_evaluation_answers_the_per_quadruple_calls_from_one_batch_per_timestamp():
    """RENet.lookahead_eval: the loop of test.py:104-139 / train.py:160-172 (one evaluate_filter call per quadruple, the whole
    stream passed as all_triplets) over two timestamps, with and without the look-ahead table: same ranks and losses, the
    same predicted graph, ONE batched evaluation per timestamp; the table is dropped when parameters change."""
    undo = EMU.install()
    try:
        import global_model as GM
        num_k = 16
        net, cfg = _setup(num_k, 20.0)
        gnet = GM.RENet_global(cfg['num_ent'], 100, cfg['num_rels'], dropout=0.0, seq_len=5, num_k=num_k, maxpool=1)
        gnet.load_state_dict({k: torch.from_numpy(v) for k, v in
                              fixtures.make_params(12, global_shapes(cfg['num_ent'], cfg['num_rels'], 100)).items()})
        gnet.eval()
        _, tr, va, te = fixtures.split_dataset('small')
        import preprocess as P
        allq = np.concatenate((tr, va, te))
        hs, ho = P.HistoryIndex(allq, 's', 5), P.HistoryIndex(allq, 'o', 5)
        idx = np.arange(len(tr), len(tr) + len(va))
        (vs, vst), (vo, vot) = hs.to_lists(idx), ho.to_lists(idx)
        total = torch.from_numpy(allq)
        valid = torch.from_numpy(va)
        t_first = int(va[0, 3])
        n_eval = int(np.count_nonzero(va[:, 3] == t_first)) + 5          # into the second timestamp: one advance
        out = []
        for look in (False, True):
            m = copy.deepcopy(net)
            m.lookahead_eval = look
            calls = []
            real = m.evaluate_filter_batch
            m.evaluate_filter_batch = lambda *a, real=real, calls=calls, **k: (calls.append(len(a[0])), real(*a, **k))[1]
            torch.manual_seed(5)
            with torch.no_grad():
                res = [m.evaluate_filter(valid[i], (vs[i], vst[i]), (vo[i], vot[i]), gnet, total) for i in range(n_eval)]
            new_t = [t for t in m.graph_dict.keys() if t not in net.graph_dict]
            facts = {t: set(map(tuple, np.stack(m.graph_dict[t].global_triples(), 1).tolist())) for t in new_t}
            out.append((np.asarray([r for r, _ in res]), np.asarray([float(l) for _, l in res]), facts, calls, m))
        (r0, l0, f0, c0, _), (r1, l1, f1, c1, m1) = out
        assert c0 == [] and len(c1) == 2, (c0, c1)                          # one batched evaluation per timestamp
        assert c1[0] == len(np.unique(va[va[:, 3] == t_first], axis=0))
        assert f0.keys() == f1.keys() and all(f0[t] == f1[t] for t in f0) and len(f0) == 1
        np.testing.assert_allclose(l1, l0, rtol=1e-5, atol=1e-5)
        assert float(np.mean(r0 == r1)) >= 0.99 and np.abs(r0 - r1).max() <= 1, (r0, r1)
        # a parameter update drops the table (validation passes of different epochs start at the same timestamp)
        t_now = int(va[n_eval - 1, 3])
        with torch.no_grad():
            m1.linear.bias.add_(0.25)
            m1.evaluate_filter(valid[n_eval - 1], (vs[n_eval - 1], vst[n_eval - 1]), (vo[n_eval - 1], vot[n_eval - 1]), gnet, total)
        assert len(c1) == 3 and m1._la['t'] == t_now
        # an empty GIVEN history with a non-empty rolling window: the plain path (zero state on that side), not the table
        k = n_eval - 1
        s_k = int(va[k, 0])
        if len(m1.s_hist_test[s_k]) != 0:
            with torch.no_grad():
                a = m1.evaluate_filter(valid[k], ([], []), (vo[k], vot[k]), gnet, total)
                m1.lookahead_eval = False
                b = m1.evaluate_filter(valid[k], ([], []), (vo[k], vot[k]), gnet, total)
            assert np.array_equal(a[0], b[0]) and float(a[1]) == float(b[1])
    finally:
        undo()
