"""CPU tests of the host side of the product: C-ABI library exports, the vectorised batch builder
against the oracle's restatement of utils.py:209-283, utilities against the reference where present."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from helpers import O, fixtures, ROOT

import graph as G
import utils as U


def _header_symbols():
    txt = open(os.path.join(ROOT, 'include', 'renet_hip.h')).read()
    return sorted(set(re.findall(r'\b(renet_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    import renet_hip
    lib_path = renet_hip.LIB_PATH
    assert os.path.isfile(lib_path), 'librenet_hip.so missing: run python re-net_amd/build.py'
    lib = ctypes.CDLL(lib_path)
    declared = _header_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), 'symbol %s declared in include/renet_hip.h but not exported' % name
    assert sorted(renet_hip.EXPORTS) == declared, 'python binding and header disagree'
    assert lib.renet_version() == 1


def test_missing_library_fails_loudly(monkeypatch):
    import renet_hip
    monkeypatch.setattr(renet_hip, '_lib', None)
    monkeypatch.setattr(renet_hip, 'LIB_PATH', '/nonexistent/librenet_hip.so')
    with pytest.raises(renet_hip.RenetHipError):
        renet_hip.lib()


def _keys(slot_t, ent):
    return [(int(t), int(e)) for t, e in zip(slot_t, ent)]


@pytest.mark.parametrize('name,lo,hi,sort', [('small', 300, 420, True), ('small', 0, 64, True),
                                              ('tiny', 60, 100, True), ('small', 500, 510, True)])
def test_batch_builder_matches_oracle(name, lo, hi, sort):
    cfg, tr, va, te = fixtures.split_dataset(name)
    idx = np.arange(lo, min(hi, len(tr)))
    _check_builder_against_oracle(tr, cfg['num_ent'], cfg['num_rels'], idx, sort)


def test_batch_builder_matches_oracle_on_random_degenerate_streams():
    """Tiny random streams with everything the fixtures rarely hold: self-loop facts (s == o), exact duplicate
    facts inside a timestamp, one-fact timestamps, entities that appear only once, batches that are all empty."""
    for seed in range(12):
        rng = np.random.RandomState(100 + seed)
        ne, nr, nt = int(rng.randint(3, 9)), int(rng.randint(1, 4)), int(rng.randint(3, 9))
        rows = []
        for t in range(nt):
            k = int(rng.randint(1, 7))
            q = np.stack((rng.randint(0, ne, k), rng.randint(0, nr, k), rng.randint(0, ne, k), np.full(k, t * 24)), 1)
            if rng.rand() < 0.5:
                q[0, 2] = q[0, 0]                                   # self loop
            if k > 1 and rng.rand() < 0.5:
                q[1] = q[0]                                         # duplicate fact
            rows.append(q)
        tr = np.concatenate(rows).astype(np.int64)
        n = len(tr)
        for idx in (np.arange(n), np.arange(min(2, n)), np.arange(n - min(3, n), n)):
            _check_builder_against_oracle(tr, ne, nr, idx, True)


def _check_builder_against_oracle(tr, num_ent, num_rels, idx, sort, history_len=10):
    from collections import Counter
    cfg = {'num_ent': num_ent, 'num_rels': num_rels}
    gd = U.build_graph_dict(tr, cfg['num_rels'])
    ogd = O.build_graph_dict(tr, cfg['num_rels'])
    (sh, sht), _, _ = O.build_histories(tr, cfg['num_ent'], history_len=history_len)
    hist, hist_t = [sh[i] for i in idx], [sht[i] for i in idx]
    fh = G.FlatHistory.from_lists(hist, hist_t)
    hb = G.build_batch(G.store_for(gd), cfg['num_ent'], cfg['num_rels'], tr[idx, 0], tr[idx, 1], fh, sort=sort)
    bg = O.batch_for_histories(hist, hist_t, tr[idx, 0], ogd, sort=sort)
    assert (hb.N, hb.E, hb.S, hb.nnz) == (bg.num_nodes, len(bg.src), len(bg.subj_row), len(bg.lens))
    assert np.array_equal(hb.perm, bg.perm) and np.array_equal(hb.lens, bg.lens)
    if hb.N == 0:
        return
    ot = np.repeat(np.asarray(bg.graph_t), np.diff(np.asarray(bg.graph_off + [bg.num_nodes])))
    okeys = _keys(ot, bg.ent)
    slot_t = hb.graph_t[hb.node_slot]
    # rows read after the last RGCN layer (the subject rows) are numbered first
    assert hb.nA == len(np.unique(hb.subj_row)) and hb.subj_row.max() < hb.nA
    m2 = np.repeat(np.arange(hb.N), np.diff(hb.row_ptr)) < hb.nA
    assert len(hb.e_src2) == int(m2.sum()) and np.all(hb.e_dst2 < hb.nA) and hb.chunk_ptr2[-1] == len(hb.e_src2)
    mkeys = _keys(slot_t, hb.node_ent)
    assert set(mkeys) == set(okeys) and len(set(mkeys)) == hb.N
    oe = Counter((okeys[a], okeys[b], int(t)) for a, b, t in zip(bg.src, bg.dst, bg.type_s))
    dst = np.repeat(np.arange(hb.N), np.diff(hb.row_ptr))
    me = Counter((mkeys[a], mkeys[b], int(t)) for a, b, t in zip(hb.col, dst, hb.etype))
    assert oe == me
    # the relation-bucketed list holds the same multiset, sorted by type, chunked within a type
    me2 = Counter((mkeys[a], mkeys[b]) for a, b in zip(hb.e_src, hb.e_dst))
    assert me2 == Counter((k[0], k[1]) for k in me.elements())
    assert hb.chunk_ptr[0] == 0 and hb.chunk_ptr[-1] == hb.E and np.all(np.diff(hb.chunk_ptr) <= G.CHUNK)
    et_sorted = np.sort(hb.etype)
    for c in range(hb.n_chunks):
        assert np.all(et_sorted[hb.chunk_ptr[c]:hb.chunk_ptr[c + 1]] == hb.chunk_type[c])
    onorm = dict(zip(okeys, bg.norm))
    assert all(abs(onorm[k] - hb.norm[i]) < 1e-7 for i, k in enumerate(mkeys))
    assert [mkeys[i] for i in hb.subj_row_seqmajor] == [okeys[i] for i in bg.subj_row]
    # packed layout: time-major, batch sizes non-increasing, row p = off[j] + i
    assert np.all(np.diff(hb.batch_sizes) <= 0) and hb.batch_sizes.sum() == hb.S
    assert np.array_equal(hb.step_t_packed[np.argsort(hb.packed_from_seqmajor, kind='stable')], bg.step_t)
    # every paired edge exists: (u -> v, t) <=> (v -> u, (t + R) mod 2R)   [what backward-wrt-h relies on]
    R = cfg['num_rels']
    assert Counter((b, a, (t + R) % (2 * R)) for (a, b, t) in me.elements()) == me


def test_flat_history_roundtrip_and_take():
    cfg, tr, va, te = fixtures.split_dataset('small')
    (sh, sht), _, _ = O.build_histories(tr, cfg['num_ent'])
    fh = G.FlatHistory.from_lists(sh, sht)
    sub = fh.take(np.array([400, 3, 250]))
    ref = G.FlatHistory.from_lists([sh[400], sh[3], sh[250]], [sht[400], sht[3], sht[250]])
    for f in ('seq_ptr', 'step_t', 'nbr_ptr', 'nbr_o'):
        assert np.array_equal(getattr(sub, f), getattr(ref, f))


def test_segplan():
    p = G.SegPlan.host(np.array([5, 2, 5, 9, 2, 2]))
    assert p.target.tolist() == [2, 5, 9] and p.seg_ptr.tolist() == [0, 3, 5, 6]
    assert p.order.tolist() == [1, 4, 5, 0, 2, 3]


@pytest.mark.reference
def test_utils_match_reference(tmp_path):
    from oracle import ref_loader
    if not ref_loader.available():
        pytest.skip('reference tree not present')
    ref = ref_loader.load()
    cfg, tr, va, te = fixtures.split_dataset('small')
    a, b = ref.utils.get_true_distribution(tr, cfg['num_ent'])
    c, d = U.get_true_distribution(tr, cfg['num_ent'])
    assert np.array_equal(a, c) and np.array_equal(b, d)
    for name, q in (('train.txt', tr), ('valid.txt', va)):
        with open(tmp_path / name, 'w') as f:
            for s, r, o, t in q:
                f.write('%d\t%d\t%d\t%d\t0\n' % (s, r, o, t))
    q1, t1 = ref.utils.load_quadruples(str(tmp_path), 'train.txt', 'valid.txt')
    q2, t2 = U.load_quadruples(str(tmp_path), 'train.txt', 'valid.txt')
    assert np.array_equal(q1, q2) and np.array_equal(t1, t2)
    with ref_loader.cpu_mode():
        g1 = ref.utils.get_big_graph(tr[tr[:, 3] == tr[0, 3]][:, :3], cfg['num_rels'])
    g2 = U.get_big_graph(tr[tr[:, 3] == tr[0, 3]][:, :3], cfg['num_rels'])
    src, dst, et = g2.edges(False)
    assert np.array_equal(g1._src.numpy(), src) and np.array_equal(g1._dst.numpy(), dst)
    assert np.array_equal(g1.edata['type_s'].numpy(), et) and np.array_equal(g1.edata['type_o'].numpy(), g2.edges(True)[2])
    assert g1.ids == g2.ids


@pytest.mark.parametrize('name', ['tiny', 'small'])
def test_history_index_matches_oracle_streaming_builder(name):
    """preprocess.HistoryIndex (vectorised snapshot ranges) vs the restated reference loop."""
    import preprocess as P
    cfg, tr, va, te = fixtures.split_dataset(name)
    allq = np.concatenate((tr, va, te))
    state = None
    ref = {'s': ([], []), 'o': ([], [])}
    for q in (tr, va, te):
        (sh, sht), (oh, oht), state = O.build_histories(q, cfg['num_ent'], state=state)
        ref['s'][0].extend(sh); ref['s'][1].extend(sht)
        ref['o'][0].extend(oh); ref['o'][1].extend(oht)
    for role in ('s', 'o'):
        hi = P.HistoryIndex(allq, role, history_len=10)
        mine = hi.to_lists(np.arange(len(allq)))
        assert fixtures.histories_equal(mine, ref[role]), role
        idx = np.array([len(allq) - 1, 5, len(allq) // 2, 0])
        a = hi.take(idx)
        b = G.FlatHistory.from_lists([ref[role][0][i] for i in idx], [ref[role][1][i] for i in idx])
        for f in ('seq_ptr', 'step_t', 'nbr_ptr', 'nbr_o'):
            assert np.array_equal(getattr(a, f), getattr(b, f))
        c = hi.take(idx, max_len=3)
        d = G.FlatHistory.from_lists([ref[role][0][i][-3:] for i in idx], [ref[role][1][i][-3:] for i in idx])
        for f in ('seq_ptr', 'step_t', 'nbr_ptr', 'nbr_o'):
            assert np.array_equal(getattr(c, f), getattr(d, f))


def test_synthetic_stream_shape():
    import synth
    q, ne, nr, unit = synth.make_stream('ICEWS18', seed=999, num_t=12)
    assert ne == 23033 and nr == 256 and unit == 24
    assert np.all(np.diff(q[:, 3]) >= 0) and len(np.unique(q[:, 3])) == 12
    assert q[:, 0].max() < ne and q[:, 1].max() < nr and 1000 < len(q) / 12 < 2200
    q2, _, _, _ = synth.make_stream('ICEWS18', seed=999, num_t=12)
    assert np.array_equal(q, q2)


@pytest.mark.reference
def test_api_surface_matches_reference_classes():
    """Same class names, constructor / method signatures (parameter names, order, defaults) as the
    reference modules the drivers import -- what makes train.py / test.py run unchanged on top of us."""
    import inspect
    from oracle import ref_loader
    if not ref_loader.available():
        pytest.skip('reference tree not present')
    ref = ref_loader.load()
    import Aggregator as A
    import RGCN as Rg
    import global_model as GM
    import model as M
    pairs = [(ref.model.RENet, M.RENet, ['__init__', 'forward', 'init_history', 'pred_r_rank2', 'predict',
                                        'evaluate', 'evaluate_filter', 'update_cache']),
             (ref.global_model.RENet_global, GM.RENet_global, ['__init__', 'forward', 'predict', 'get_global_emb',
                                                               'update_global_emb']),
             (ref.Aggregator.RGCNAggregator, A.RGCNAggregator, ['__init__', 'forward', 'predict_batch', 'predict']),
             (ref.Aggregator.RGCNAggregator_global, A.RGCNAggregator_global, ['__init__', 'forward', 'predict']),
             (ref.RGCN.RGCNBlockLayer, Rg.RGCNBlockLayer, ['__init__', 'forward']),
             (ref.RGCN.RGCNLayer, Rg.RGCNLayer, ['__init__'])]
    for rc, mc, names in pairs:
        for n in names:
            rs, ms = inspect.signature(getattr(rc, n)), inspect.signature(getattr(mc, n))
            rp = [(p.name, p.default) for p in rs.parameters.values()]
            mp = [(p.name, p.default) for p in ms.parameters.values()]
            assert rp == mp, (rc.__name__, n, rp, mp)
    for fn in ('get_total_number', 'load_quadruples', 'make_batch', 'make_batch2', 'get_big_graph', 'get_data',
               'get_true_distribution', 'soft_cross_entropy', 'cuda', 'move_dgl_to_cuda'):
        rs = inspect.signature(getattr(ref.utils, fn))
        ms = inspect.signature(getattr(U, fn))
        assert list(rs.parameters) == list(ms.parameters), fn
    # state_dict keys and shapes (checkpoint compatibility)
    r_net = ref.model.RENet(50, 200, 7, dropout=0.5, seq_len=10, num_k=10)
    m_net = M.RENet(50, 200, 7, dropout=0.5, seq_len=10, num_k=10)
    assert {k: tuple(v.shape) for k, v in r_net.state_dict().items()} == \
        {k: tuple(v.shape) for k, v in m_net.state_dict().items()}
    r_g = ref.global_model.RENet_global(50, 200, 7, dropout=0.5, seq_len=10, num_k=10, maxpool=1)
    m_g = GM.RENet_global(50, 200, 7, dropout=0.5, seq_len=10, num_k=10, maxpool=1)
    assert {k: tuple(v.shape) for k, v in r_g.state_dict().items()} == \
        {k: tuple(v.shape) for k, v in m_g.state_dict().items()}


def test_preprocess_cli_writes_reference_layout(tmp_path):
    import pickle
    import preprocess as P
    cfg, tr, va, te = fixtures.split_dataset('tiny')
    for name, q in (('train.txt', tr), ('valid.txt', va), ('test.txt', te)):
        with open(tmp_path / name, 'w') as f:
            for s, r, o, t in q:
                f.write('%d\t%d\t%d\t%d\t0\n' % (s, r, o, t))
    with open(tmp_path / 'stat.txt', 'w') as f:
        f.write('%d\t%d\t0\n' % (cfg['num_ent'], cfg['num_rels']))
    P.write_reference_pickles(str(tmp_path))
    gold = np.load(os.path.join(ROOT, 'tests', 'golden', 'prep_tiny.npz'))
    for split, fn in (('train', 'train'), ('valid', 'dev'), ('test', 'test')):
        for tag, name in (('s', 'sub'), ('o', 'ob')):
            with open(tmp_path / ('%s_history_%s.txt' % (fn, name)), 'rb') as f:
                mine = pickle.load(f)
            ref = fixtures.unflatten_histories(gold['%s_%s_seq_ptr' % (split, tag)], gold['%s_%s_step_t' % (split, tag)],
                                               gold['%s_%s_nbr_ptr' % (split, tag)], gold['%s_%s_nbr' % (split, tag)])
            assert fixtures.histories_equal((mine[0], mine[1]), ref), (split, tag)
    with open(tmp_path / 'train_graphs.txt', 'rb') as f:
        gd = pickle.load(f)
    assert list(gd.keys()) == gold['graph_t'].tolist()
    k = len(gd) // 2
    t = list(gd.keys())[k]
    e0, e1 = gold['graph_edge_ptr'][k], gold['graph_edge_ptr'][k + 1]
    src, dst, et = gd[t].edges(False)
    assert np.array_equal(src, gold['graph_src'][e0:e1]) and np.array_equal(et, gold['graph_type_s'][e0:e1])


def test_prefetcher_returns_packed_batches_in_order():
    """pipeline.BatchPrefetcher (forked workers) == inline builder, same order; PackedBatch round trip."""
    import pickle
    import pipeline
    import preprocess as P
    import synth
    quads, ne, nr, _ = synth.make_stream('ICEWS18', seed=1, num_t=20)
    gd = P.build_graph_dict(quads, nr)
    hs = P.HistoryIndex(quads, 's', 10)
    perm = np.random.RandomState(0).permutation(len(quads))
    store = G.store_for(gd)

    def fn(step):
        idx = perm[step * 128:(step + 1) * 128]
        return G.PackedBatch(G.build_batch(store, ne, nr, quads[idx, 0], quads[idx, 1], hs.take(idx), sort=True))
    inline = [fn(k) for k in range(5)]
    piped = list(pipeline.BatchPrefetcher(fn, range(5), 3))
    for x, y in zip(inline, piped):
        assert np.array_equal(x.ints, y.ints) and np.array_equal(x.norm, y.norm) and x.scalars == y.scalars
        assert x.names == y.names and x.offs == y.offs and np.array_equal(x.host_small['perm'], y.host_small['perm'])
    z = pickle.loads(pickle.dumps(inline[0], protocol=5))
    assert np.array_equal(z.ints, inline[0].ints) and all(o % 4 == 0 for o in z.offs)


def test_native_host_builder_equals_numpy_builder():
    """csrc/host_builder.cpp (edge filter, CSR / bucket layouts, plans) vs the numpy specification: every
    array of the packed batch identical, for a training-shaped and a tiny batch."""
    import preprocess as P
    import synth
    quads, ne, nr, _ = synth.make_stream('ICEWS18', seed=3, num_t=40)
    gd = P.build_graph_dict(quads, nr)
    hs = P.HistoryIndex(quads, 'o', 10)
    perm = np.random.RandomState(2).permutation(len(quads))
    store = G.store_for(gd)
    for nb in (700, 5):
        idx = perm[:nb]
        packed = {}
        for native in (False, True):
            G.NATIVE = native
            try:
                hb = G.build_batch(store, ne, nr, quads[idx, 2], quads[idx, 1], hs.take(idx), sort=True)
                packed[native] = (G.PackedBatch(hb), hb)
            finally:
                G.NATIVE = True
        a, b = packed[False][0], packed[True][0]
        assert a.names == b.names and a.sizes == b.sizes and a.scalars == b.scalars
        assert np.array_equal(a.ints, b.ints) and np.array_equal(a.norm, b.norm)
        assert np.array_equal(packed[False][1].heavy_rows, packed[True][1].heavy_rows)
    # full-graph batches (global model) and explicit edge lists go through the same native pass
    for native in (False, True):
        G.NATIVE = native
        try:
            packed[native] = G.PackedBatch(G.build_full_graphs(gd, list(gd.keys())[:12]))
        finally:
            G.NATIVE = True
    assert np.array_equal(packed[False].ints, packed[True].ints) and np.array_equal(packed[False].norm, packed[True].norm)


def _subject_row_signature(hb):
    """For every packed step row: (sequence in the caller's order, step) -> (norm of the subject row, sorted
    multiset of its in-edges as (source entity, relation type))."""
    sig = {}
    step_of_row = np.arange(hb.S) - hb.step_off[np.searchsorted(hb.step_off, np.arange(hb.S), side='right') - 1]
    j_of_row = np.searchsorted(hb.step_off, np.arange(hb.S), side='right') - 1
    del step_of_row
    for p in range(hb.S):
        row = int(hb.subj_row[p])
        lo, hi = int(hb.row_ptr[row]), int(hb.row_ptr[row + 1])
        edges = sorted(zip(hb.node_ent[hb.col[lo:hi]].tolist(), hb.etype[lo:hi].tolist()))
        sig[(int(hb.perm[hb.row_seq[p]]), int(j_of_row[p]))] = (float(hb.norm[row]), edges)
    return sig


def test_grouped_batch_equals_one_call_per_group():
    """build_batch(group=...) = the member graphs of calling the builder once per group (the reference's
    inference calls get_s_r_embed_rgcn per test quadruple, model.py:329-352), for both builder back ends, with
    more (group, timestamp) slots than one lookup-table chunk holds."""
    import preprocess as P
    import synth
    quads, ne, nr, _ = synth.make_stream('ICEWS18', seed=5, num_t=30)
    gd = P.build_graph_dict(quads, nr)
    hs = P.HistoryIndex(quads, 's', 10)
    idx = np.random.RandomState(4).permutation(len(quads))[:120]
    store = G.store_for(gd)
    s, r, fh = quads[idx, 0], quads[idx, 1], hs.take(idx)
    group = np.arange(len(idx)) // 2                     # pairs of sequences share their member graphs
    sigs = []
    for native, entries in ((False, 1 << 24), (True, 1 << 24), (True, 37 * ne), (False, 37 * ne)):
        # numpy: dense scan in chunks of 37 slots; native: the sparse per-node walk in chunks of ~3 timestamps' facts
        G.NATIVE, G.TABLE_ENTRIES, G.SPARSE_FACTS = native, entries, (1 << 23 if entries == 1 << 24 else 5000)
        try:
            sigs.append(_subject_row_signature(G.build_batch(store, ne, nr, s, r, fh, sort=True, group=group)))
        finally:
            G.NATIVE, G.TABLE_ENTRIES, G.SPARSE_FACTS = True, 1 << 24, 1 << 23
    assert sigs[0] == sigs[1] == sigs[2] == sigs[3]
    want = {}
    for g in np.unique(group):
        m = np.nonzero(group == g)[0]
        hb = G.build_batch(store, ne, nr, s[m], r[m], fh.take(m), sort=True)
        for (i, j), v in _subject_row_signature(hb).items():
            want[(int(m[i]), j)] = v
    assert sigs[1] == want
    # the default (training) semantics differ: node sets are unions over the whole batch
    union = _subject_row_signature(G.build_batch(store, ne, nr, s, r, fh, sort=True))
    assert union.keys() == want.keys() and union != want


def test_ctypes_signatures_match_the_header():
    """Every prototype of include/renet_hip.h against the argtypes / restype renet_hip.py binds it with: same
    arity, same class per argument.  (ctypes accepts EXTRA positional arguments and converts them as 32-bit ints:
    a binding two entries short once passed the stream handle that way, leaving the upper half of the pointer to
    whatever was on the stack.)"""
    import renet_hip as K
    hdr = re.sub(r'/\*.*?\*/', '', open(os.path.join(ROOT, 'include', 'renet_hip.h')).read(), flags=re.S)
    protos = re.findall(r'\b(int|size_t|void|int64_t)\s+(renet_\w+)\s*\(([^)]*)\)\s*;', hdr)
    assert len(protos) >= 30
    kinds = {'int': 'i32', 'int32_t': 'i32', 'float': 'f32', 'uint64_t': 'u64', 'size_t': 'size', 'int64_t': 'i64'}
    ckind = {ctypes.c_int: 'i32', ctypes.c_int32: 'i32', ctypes.c_float: 'f32', ctypes.c_uint64: 'u64',
             ctypes.c_size_t: 'size', ctypes.c_int64: 'i64', ctypes.c_void_p: 'ptr', ctypes.c_ulong: 'u64',
             ctypes.c_long: 'i64'}
    for ret, name, params in protos:
        want = []
        for prm in [x.strip() for x in params.split(',') if x.strip() and x.strip() != 'void']:
            want.append('ptr' if '*' in prm else kinds[prm.rsplit(' ', 1)[0].replace('const', '').strip()])
        assert name in K._SIGNATURES, 'no binding for ' + name
        restype, argtypes = K._SIGNATURES[name]
        got = [ckind[a] for a in argtypes]
        got = ['size' if (g == 'u64' and w == 'size') else g for g, w in zip(got, want)] + got[len(want):]
        assert got == want, (name, got, want)
        assert (restype is None) == (ret == 'void') and (ret == 'void' or ckind[restype] in (kinds[ret], 'u64'))


def _tile_of_block(L, nbx, nby, panel=8, nbz=1):
    """Python restatement of tile_of_block (csrc/gemm_split.hip): linear dispatch index (x fastest, then y, then z) ->
    (bx, by) for nbz == 1, (bx, by, bz) for split-K grids."""
    nb = nbx * nby
    if nbz == 1:
        per = nb >> 3
        t = (L & 7) * per + (L >> 3) if L < 8 * per else L
        bz = 0
    else:
        per3 = (nb * nbz) >> 3
        v = (L & 7) * per3 + (L >> 3) if L < 8 * per3 else L
        bz, t = divmod(v, nb)
    ns, nl = min(nbx, nby), max(nbx, nby)
    w = min(ns, panel)
    p, r = divmod(t, w * nl)
    wp = min(w, ns - p * w)
    l, sh = r // wp, p * w + r % wp
    xy = (l, sh) if nby <= nbx else (sh, l)
    return xy if nbz == 1 else xy + (bz,)


def test_gemm_virtual_tile_order_is_a_bijection():
    """The XCD-aware tile order of the bf16x6 GEMM must visit every tile exactly once for every grid shape (the
    formula here mirrors the device code line by line; the GPU tests cover a handful of shapes, this covers all
    small ones), and consecutive workgroups of one XCD must share a slab of the long operand."""
    src = open(os.path.join(ROOT, 're-net_amd', 'csrc', 'gemm_split.hip')).read()
    for frag in ('(L & 7) * per + (L >> 3)', 'const int w = min(ns, xcd_order);', 'const int wp = min(w, ns - p * w);',
                 'const int l = r / wp, sh = p * w + (r - l * wp);', '(L3 & 7) * per3 + (L3 >> 3)',
                 'const int L3 = blockIdx.x + nbx * (blockIdx.y + nby * blockIdx.z);', 'bz = v / nb;'):
        assert frag in src, 'tile_of_block changed: update the restatement in this test (%s)' % frag
    for nbx in list(range(1, 41)) + [180, 181, 360]:
        for nby in list(range(1, 41)) + [8, 180]:
            for panel in ((8, 4, 2) if nbx > 24 or nby > 24 else (1, 2, 3, 4, 8, 16)):   # (round 4: the panel width is a knob,
                                                                                       # chosen per shape by panel_width())
                seen = {_tile_of_block(L, nbx, nby, panel) for L in range(nbx * nby)}
                assert len(seen) == nbx * nby and all(0 <= x < nbx and 0 <= y < nby for x, y in seen), (nbx, nby, panel)
    # logits GEMM of the bench (8 x 180 tiles): the 8 row tiles of one column tile are consecutive on one XCD
    nbx, nby = 180, 8
    xcd0 = [_tile_of_block(L, nbx, nby) for L in range(0, nbx * nby, 8)]
    assert [t[0] for t in xcd0[:16]] == [0] * 8 + [1] * 8 and [t[1] for t in xcd0[:8]] == list(range(8))
    # split-K grids: a bijection over (x, y, z), and every XCD (flattened index mod 8) touches as few k-slices as an
    # eighth of the sequence can: the weight-gradient GEMMs of the bench (5 x 7 x 14, 5 x 2 x 39), dfeat (5 x 8 x 6)
    for nbx, nby, nbz in [(7, 5, 14), (2, 5, 39), (5, 8, 6), (1, 1, 9), (3, 3, 2), (5, 5, 20), (2, 2, 82), (4, 2, 15)]:
        n = nbx * nby * nbz
        seen = {_tile_of_block(L, nbx, nby, 8, nbz) for L in range(n)}
        assert len(seen) == n and all(0 <= x < nbx and 0 <= y < nby and 0 <= z < nbz for x, y, z in seen), (nbx, nby, nbz)
        for xcd in range(8):
            zs = {_tile_of_block(L, nbx, nby, 8, nbz)[2] for L in range(xcd, n, 8)}
            # a contiguous eighth may straddle one slice boundary more, and the last (n % 8) blocks keep their own index
            assert len(zs) <= -(-nbz // 8) + 2, (nbx, nby, nbz, xcd, sorted(zs))


def test_gemm_dispatch_plan_of_the_bench_shapes():
    """renet_gemm_split_plan: the launcher's own decision function (csrc/gemm_split.hip plan_split), queried without a GPU.
    Pins what DESIGN / profiles say the step's GEMMs run on: kernel family, loader, tile order, grid."""
    import renet_hip as K
    if os.environ.get('RENET_GEMM_TILE_ORDER') or os.environ.get('RENET_GEMM_PANEL_W') or os.environ.get('RENET_GEMM_TALL') or os.environ.get('RENET_GEMM_KERNEL'):
        pytest.skip('dispatch knobs set in the environment')
    S = 15439
    logits = K.gemm_split_plan(0, 1, 2048, 23033, 600)
    assert logits == {'kernel': 'two_phase_256', 'raw': True, 'xcd_order': 4, 'grid': (180, 8, 1), 'split_k': 1}
    dfeat = K.gemm_split_plan(0, 0, 2048, 600, 23033, split_k=6)
    assert dfeat['kernel'] == 'two_phase_256' and dfeat['grid'] == (5, 8, 6) and dfeat['xcd_order'] == 8
    dw = K.gemm_split_plan(1, 0, 23033, 600, 2048)              # long operand 189 MB: panels stay 8 wide
    assert dw['kernel'] == 'two_phase_128' and dw['grid'] == (5, 180, 1) and dw['xcd_order'] == 8 and dw['raw']
    proj = K.gemm_split_plan(0, 1, S, 600, 800)                 # 2 MB short operand: fits an L2, no narrowing
    assert proj['kernel'] == 'two_phase_128' and proj['xcd_order'] == 8 and proj['grid'] == (5, 121, 1)
    dwih = K.gemm_split_plan(1, 0, 600, 800, S, split_k=14)
    assert dwih['kernel'] == 'two_phase_128' and dwih['grid'] == (7, 5, 14)
    assert K.gemm_split_plan(0, 1, 23033, 200, 200)['kernel'] == 'weight_resident'
    assert K.gemm_split_plan(0, 0, 2048, 400, 256)['kernel'] == 'fused'                     # <= 256 tiles
    assert K.gemm_split_plan(1, 0, 256, 400, 2048, split_k=15)['kernel'] == 'fused'
    assert K.gemm_split_plan(0, 1, 4096, 4096, 4096)['xcd_order'] == 8                      # 67 MB short operand: re-reading
    # the long one per extra panel would cost more than the sweep
    big = K.gemm_split_plan(0, 1, 1 << 20, 600, 2048)           # A reaches 2^31 elements: 64-bit loader
    assert big['raw'] is False and big['kernel'] in ('two_phase_128', 'two_phase_256')
    assert K.gemm_split_plan(0, 1, 300, 300, 64, split_k=9)['split_k'] == 2                 # clamped to the k-tiles
    with pytest.raises(K.RenetHipError):
        K.gemm_split_plan(0, 0, 0, 5, 5)


def test_split_k_cost_model():
    import renet_hip as K
    assert K.auto_split_k(1024, 23033, 600) == 1 and K.auto_split_k(23033, 600, 1024) == 1     # full grids
    assert K.auto_split_k(7624, 600, 800) == 1
    assert 8 <= K.auto_split_k(1024, 600, 23033) <= 16                                          # swept optimum: 12
    assert 10 <= K.auto_split_k(600, 800, 7624) <= 20                                           # swept optimum: 14
    assert K.auto_split_k(200, 200, 46075) >= 64
    assert K.auto_split_k(100, 100, 46075) >= 64          # a single tile must still be split (n_hidden = 100)
    assert K.auto_split_k(64, 64, 96) == 1                # too few k-tiles to split
    for m, n, k in ((1, 1, 1), (128, 128, 32), (300, 100, 7624), (1200, 400, 8000)):
        s = K.auto_split_k(m, n, k)
        assert 1 <= s <= max(1, ((k + 31) // 32))


def test_graph_store_cache_survives_address_reuse():
    """store_for() recognises a graph_dict by object identities; graphs that were freed and re-created at the same
    addresses (same-shaped dicts built in a loop) must not resurrect a stale store."""
    import gc
    import preprocess as P
    seen = []
    for k in range(6):
        rng = np.random.RandomState(k)
        q = np.stack((rng.randint(0, 9, 40), rng.randint(0, 3, 40), rng.randint(0, 9, 40), np.repeat(np.arange(4), 10) * 24), 1)
        gd = P.build_graph_dict(q.astype(np.int64), 3)
        st = G.store_for(gd)
        s_, r_, o_ = st.trip_s.copy(), st.trip_r.copy(), st.trip_o.copy()
        want = np.concatenate([np.stack(gd[t].global_triples(), 1) for t in gd])
        assert np.array_equal(np.stack((s_, r_, o_), 1), want), k
        seen.append(id(gd))
        del gd, st
        gc.collect()


def test_history_index_and_builder_on_hypothesis_streams():
    """Property test (hypothesis): for arbitrary small time-ordered quadruple streams -- including repeated facts,
    self loops, entities that vanish and return, history_len shorter than the stream -- HistoryIndex equals the
    restated reference loop and the batch builder equals the oracle for every window of the stream."""
    import preprocess as P
    from hypothesis import given, settings, strategies as st

    fact = st.tuples(st.integers(0, 5), st.integers(0, 2), st.integers(0, 5))
    stream = st.lists(st.lists(fact, min_size=1, max_size=5), min_size=2, max_size=7)

    @settings(max_examples=40, deadline=None)
    @given(stream, st.integers(1, 4))
    def check(per_t, hist_len):
        q = np.asarray([(s, r, o, 24 * t) for t, facts in enumerate(per_t) for (s, r, o) in facts], dtype=np.int64)
        ne, nr = 6, 3
        (sh, sht), (oh, oht), _ = O.build_histories(q, ne, history_len=hist_len)
        for role, ref in (('s', (sh, sht)), ('o', (oh, oht))):
            hi = P.HistoryIndex(q, role, history_len=hist_len)
            assert fixtures.histories_equal(hi.to_lists(np.arange(len(q))), ref), role
        _check_builder_against_oracle(q, ne, nr, np.arange(len(q)), True, history_len=hist_len)

    check()


def test_batched_filtered_ranks_equal_the_per_quadruple_rule():
    """model._known_pairs + model._rank_rows (evaluate_filter_batch) vs the restated reference rule
    (model.py:384-419: sigmoid scores, all other known-true completions zeroed, ties averaged), row by row."""
    import model as M
    rng = np.random.RandomState(3)
    ne, nr, n = 17, 4, 40
    at = np.stack((rng.randint(0, ne, 300), rng.randint(0, nr, 300), rng.randint(0, ne, 300), rng.randint(0, 9, 300)), 1)
    tr = at[rng.choice(len(at), n, replace=False)]
    scores = torch.from_numpy(np.round(rng.randn(n, ne), 1).astype(np.float32))        # rounded: plenty of exact ties
    s, r, o = tr[:, 0], tr[:, 1], tr[:, 2]
    ro, co = M._known_pairs(torch.from_numpy(at), (0, 1), 2, np.stack((s, r), 1))
    ranks = M._rank_rows(scores.clone(), torch.from_numpy(o), torch.from_numpy(ro), torch.from_numpy(co))
    for i in range(n):
        known = at[(at[:, 0] == s[i]) & (at[:, 1] == r[i]), 2]
        want = O.filtered_rank(scores[i].clone(), int(o[i]), torch.from_numpy(known))
        assert ranks[i] == want, (i, ranks[i], want)
    rs, cs = M._known_pairs(at, (2, 1), 0, np.stack((o, r), 1))
    ranks_s = M._rank_rows(scores.clone(), torch.from_numpy(s), torch.from_numpy(rs), torch.from_numpy(cs))
    for i in range(n):
        known = at[(at[:, 2] == o[i]) & (at[:, 1] == r[i]), 0]
        assert ranks_s[i] == O.filtered_rank(scores[i].clone(), int(s[i]), torch.from_numpy(known))
    raw = M._rank_rows(scores.clone(), torch.from_numpy(o))                             # raw (unfiltered) variant
    for i in range(n):
        g = scores[i, o[i]]
        assert raw[i] == float((scores[i] > g).sum()) + (float((scores[i] == g).sum()) - 1.0) / 2 + 1


@pytest.mark.parametrize('heavy,budget', [(24, 32), (8, 16), (0, 1), (40, 23), (62, 1)])
def test_gather_item_plan_covers_every_light_row_once(heavy, budget):
    """graph.plan_gather_items (numpy specification) and csrc/host_builder.cpp:renet_host_gather_items: the item
    stream replays the CSR of the light rows exactly (edges in row order, then the row's flush item), hub rows are
    absent, every group holds <= 64 items, the groups of rows < n_out are a prefix and no group straddles n_out."""
    rng = np.random.RandomState(heavy * 100 + budget)
    n = 700
    deg = np.minimum(rng.zipf(1.6, n) - 1, 300)                   # many 0/1/2, a Zipf tail of hubs
    deg[rng.randint(0, n, 40)] = 0
    dst = np.repeat(np.arange(n), deg)
    src = rng.randint(0, n, len(dst))
    et = rng.randint(0, 14, len(dst))
    n_out = 233
    plans = {}
    for native in (False, True):
        G.NATIVE = native
        try:
            hb = G.HostBatch().set_edges(n, src, dst, et, 14, heavy=heavy)
            hb.set_out_rows(n_out, src, dst, et)
            hb.set_gather_plan(n_out, heavy=heavy, budget=budget)
            plans[native] = hb
        finally:
            G.NATIVE = True
    a, b = plans[False], plans[True]
    for f in ('it_src', 'it_type', 'grp_ptr', 'heavy_rows', 'heavy_rows_out'):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    assert (a.n_groups, a.n_groups_out) == (b.n_groups, b.n_groups_out)
    hb = b
    d = np.diff(hb.row_ptr)
    assert np.array_equal(hb.heavy_rows, np.nonzero(d > heavy)[0])
    sizes = np.diff(hb.grp_ptr)
    assert len(sizes) == hb.n_groups and sizes.min() >= 1 and sizes.max() <= 64 and hb.grp_ptr[0] == 0
    assert hb.grp_ptr[-1] == len(hb.it_src) == int(d[d <= heavy].sum() + np.count_nonzero(d <= heavy))
    # replay
    seen, cur = [], []
    grp_of_row = {}
    for gi in range(hb.n_groups):
        for i in range(hb.grp_ptr[gi], hb.grp_ptr[gi + 1]):
            if hb.it_type[i] >= 0:
                cur.append((int(hb.it_src[i]), int(hb.it_type[i])))
            else:
                assert hb.it_type[i] == -1
                v = int(hb.it_src[i])
                e0, e1 = hb.row_ptr[v], hb.row_ptr[v + 1]
                assert cur == list(zip(hb.col[e0:e1].tolist(), hb.etype[e0:e1].tolist())), v
                seen.append(v)
                grp_of_row[v] = gi
                cur = []
        assert cur == []                                           # a row never straddles two groups
    assert seen == np.nonzero(d <= heavy)[0].tolist()              # each light row once, ascending
    for v, gi in grp_of_row.items():
        assert (gi < hb.n_groups_out) == (v < n_out)
    with pytest.raises(ValueError):
        G.plan_gather_items(hb.row_ptr, hb.col, hb.etype, n_out, 40, 24)


def test_builder_rejects_ids_outside_the_declared_ranges():
    """the native host passes index scratch tables with entity / relation ids unchecked -- ids outside
    [0, num_ent) / [0, num_rels) (stat.txt disagreeing with the data) must raise before any native call."""
    import preprocess as P
    import synth
    quads, ne, nr, _ = synth.make_stream('YAGO', seed=3, num_t=12)
    gd = P.build_graph_dict(quads, nr)
    hs = P.HistoryIndex(quads, 's', 10)
    idx = np.arange(len(quads) - 50, len(quads))
    store = G.store_for(gd)
    G.build_batch(store, ne, nr, quads[idx, 0], quads[idx, 1], hs.take(idx))               # fine
    with pytest.raises(ValueError):
        G.build_batch(store, int(quads[:, [0, 2]].max()), nr, quads[idx, 0], quads[idx, 1], hs.take(idx))
    with pytest.raises(ValueError):
        G.build_batch(store, ne, int(quads[:, 1].max()), quads[idx, 0], quads[idx, 1], hs.take(idx))
    bad = quads[idx, 0].copy()
    bad[0] = -1
    with pytest.raises(ValueError):
        G.build_batch(store, ne, nr, bad, quads[idx, 1], hs.take(idx))


def test_thread_prefetcher_keeps_order_and_propagates_errors():
    """pipeline.BatchPrefetcher(threads=True): results in step order whatever the completion order, bounded
    look-ahead, and a worker's exception reaches the consumer."""
    import time
    import pipeline

    def fn(s):
        time.sleep(0.002 * ((7 * s) % 5))
        return s * s
    assert list(pipeline.BatchPrefetcher(fn, range(37), 6, threads=True)) == [s * s for s in range(37)]
    assert list(pipeline.BatchPrefetcher(fn, range(3), 1, threads=True)) == [0, 1, 4]        # serial path

    def bad(s):
        if s == 5:
            raise ValueError('boom')
        return s
    got = []
    with pytest.raises(ValueError):
        for x in pipeline.BatchPrefetcher(bad, range(10), 4, threads=True):
            got.append(x)
    assert got == [0, 1, 2, 3, 4]


def test_bulk_cache_update_equals_one_update_cache_call_per_winner():
    """model._bulk_update_cache (round 4: the inference advance groups the winners of a timestamp by entity) against the
    reference-shaped sequence of update_cache calls (model.py:421-446 / 254-258): same rows in the same order, for empty
    and pre-filled caches, repeated pairs and repeated relations."""
    import model as M

    class Dummy(object):
        in_dim = 50
    rng = np.random.RandomState(0)
    for trial in range(200):
        n0 = int(rng.randint(0, 6))
        start = np.unique(rng.randint(0, 4, (n0, 2)) * np.array([1, 7]), axis=0) if n0 else []
        pairs = [(int(r), int(o)) for r, o in zip(rng.randint(0, 4, 12), rng.randint(0, 5, 12) * 7)]
        seq = start
        for r, o in pairs:
            seq = M._update_cache(Dummy(), seq, r, np.asarray([o]))
        got = M._bulk_update_cache(start, pairs)
        assert np.array_equal(np.asarray(seq, dtype=np.int64).reshape(-1, 2), got), (trial, start, pairs)


def test_gemm_mode_scopes():
    """renet_hip.gemm_mode (round 4): nesting, restoration after an exception, None = no change; 'f32' is a per-model mode
    since round 5, the storage modes ('bf16s', 'bf16') stay process-wide: refused unless they ARE the process default."""
    import renet_hip as K
    base = K.GEMM_MODE
    if base not in ('bf16x6', 'f16x3'):
        pytest.skip('process default is not a split mode')
    other = 'f16x3' if base == 'bf16x6' else 'bf16x6'
    assert K.current_mode() == base
    with K.gemm_mode(None):
        assert K.current_mode() == base
    with K.gemm_mode(other):
        assert K.current_mode() == other
        with K.gemm_mode(base):
            assert K.current_mode() == base
        assert K.current_mode() == other
    assert K.current_mode() == base
    with pytest.raises(ValueError):
        with K.gemm_mode(other):
            raise ValueError('boom')
    assert K.current_mode() == base
    with K.gemm_mode('f32'):
        assert K.current_mode() == 'f32'
    assert K.current_mode() == base
    for refused in ('bf16s', 'bf16', 'nonsense'):
        with pytest.raises(K.RenetHipError):
            with K.gemm_mode(refused):
                pass
    with K.gemm_mode(base):                      # the process default itself is always accepted
        pass


def test_c_list_flattener_equals_the_numpy_formulation():
    """csrc/listwalk.c (graph._listwalk, optional) against FlatHistory.from_lists' numpy path: real histories, empty sequences and
    steps, strided / Fortran-ordered / int32 step arrays, numpy and Python timestamps, tuples (declined -> numpy path)."""
    if G._listwalk is None:
        pytest.skip('_renet_listwalk.so not built (python re-net_amd/build.py)')
    cfg, tr, va, te = fixtures.split_dataset('small')
    (sh, sht), _, _ = O.build_histories(tr, cfg['num_ent'])
    rs = np.random.RandomState(3)
    pick = rs.permutation(len(sh))[:300].tolist()
    hist = [list(sh[i]) for i in pick] + [[], [np.zeros((0, 2), dtype=np.int64)]]
    hist_t = [[np.int64(t) if k % 2 else int(t) for k, t in enumerate(sht[i])] for i in pick] + [[], [7]]
    hist[0] = [np.asfortranarray(a) for a in hist[0]]
    hist[1] = [np.concatenate((a, a))[::2] for a in hist[1]]            # strided views
    hist[2] = [a.astype(np.int32) for a in hist[2]]                      # declined by the C path -> numpy formulation, same values

    def flat(use_c):
        saved = G._listwalk
        if not use_c:
            G._listwalk = None
        try:
            return G.FlatHistory.from_lists(hist, hist_t)
        finally:
            G._listwalk = saved
    a, b = flat(True), flat(False)
    for f in ('seq_ptr', 'step_t', 'nbr_ptr', 'nbr_o'):
        assert np.array_equal(getattr(a, f), getattr(b, f)) and getattr(a, f).dtype == np.int64, f
    hist2 = [h for k, h in enumerate(hist) if k != 2]
    hist_t2 = [h for k, h in enumerate(hist_t) if k != 2]
    lens, cnt, nbr, st = (np.frombuffer(x, dtype=np.int64) for x in G._listwalk.flatten(hist2, hist_t2))
    ref = G.FlatHistory.from_lists(tuple(hist2), tuple(hist_t2))        # tuples: the numpy path
    assert np.array_equal(np.concatenate(([0], np.cumsum(lens))), ref.seq_ptr) and np.array_equal(nbr, ref.nbr_o)
    assert np.array_equal(st, ref.step_t) and np.array_equal(np.concatenate(([0], np.cumsum(cnt))), ref.nbr_ptr)
    with pytest.raises(TypeError):
        G._listwalk.flatten([[np.zeros((2, 3), dtype=np.int64)]], [[1]])
    with pytest.raises(TypeError):
        G._listwalk.flatten([[np.zeros((2, 2), dtype=np.int64)]], [[1, 2]])
