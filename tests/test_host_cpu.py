"""CPU tests of the host side of the product: C-ABI library exports, the vectorised batch builder
against the oracle's restatement of utils.py:209-283, utilities against the reference where present."""
import ctypes
import os
import re

import numpy as np
import pytest

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
    from collections import Counter
    cfg, tr, va, te = fixtures.split_dataset(name)
    gd = U.build_graph_dict(tr, cfg['num_rels'])
    ogd = O.build_graph_dict(tr, cfg['num_rels'])
    (sh, sht), _, _ = O.build_histories(tr, cfg['num_ent'])
    idx = np.arange(lo, min(hi, len(tr)))
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
    slot_t = hb.graph_t[np.searchsorted(hb.graph_off, np.arange(hb.N), side='right') - 1]
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
