"""The merged batch of both passes of a training step (graph.build_batch_both / RENet.loss_prepared_both) on CPU:
host-side structure against two separate build_batch calls, and -- with the device wrappers emulated in torch-CPU
(tests/cpu_abi_emulation.py) -- loss and every gradient against loss_prepared(subject) + loss_prepared(object)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import graph as G
import preprocess as P
import synth


def _data(num_t=40, cap=600):
    quads, num_ent, num_rels, _ = synth.make_stream('YAGO', seed=7, num_t=num_t)
    quads = quads[(quads[:, 0] < cap) & (quads[:, 2] < cap)]
    return quads, cap, num_rels


def test_merged_batch_is_the_two_batches_side_by_side():
    quads, num_ent, R = _data()
    gd = P.build_graph_dict(quads, R)
    store = G.store_for(gd)
    hs, ho = P.HistoryIndex(quads, 's'), P.HistoryIndex(quads, 'o')
    idx = np.random.RandomState(1).permutation(len(quads))[:200]
    b = quads[idx]
    s, r, o = b[:, 0], b[:, 1], b[:, 2]
    hb_s = G.build_batch(store, num_ent, R, s, r, hs.take(idx))
    hb_o = G.build_batch(store, num_ent, R, o, r, ho.take(idx))
    hb = G.build_batch_both(store, num_ent, R, s, r, o, hs.take(idx), ho.take(idx))
    B = len(b)
    assert hb.B == 2 * B and hb.N == hb_s.N + hb_o.N and hb.E == hb_s.E + hb_o.E
    assert hb.nA == hb_s.nA + hb_o.nA and hb.S == hb_s.S + hb_o.S and hb.nnz == hb_s.nnz + hb_o.nnz
    is_obj = hb.perm >= B
    assert np.array_equal(hb.is_obj, is_obj)
    # sorted rows: entity, relation-embedding row, labels
    assert np.array_equal(hb.s_sorted, np.concatenate((s, o))[hb.perm])
    assert np.array_equal(hb.rel_label, np.concatenate((r, r))[hb.perm])
    assert np.array_equal(hb.r_sorted, hb.rel_label + R * is_obj)
    assert np.array_equal(hb.ent_label, np.concatenate((o, s))[hb.perm])
    assert np.array_equal(hb.row_rel, hb.r_sorted[hb.row_seq])
    # the stable length sort keeps each direction's own order: the merged permutation restricted to one
    # direction is that direction's permutation
    assert np.array_equal(hb.perm[~is_obj], hb_s.perm) and np.array_equal(hb.perm[is_obj] - B, hb_o.perm)

    # edges as (entity of src, entity of dst, timestamp, type) multisets per direction: the object side holds
    # type_o = (type_s + R) mod 2R of its own pass's edges (model.py:78), the subject side type_s
    col, rp, et = np.asarray(hb.col), np.asarray(hb.row_ptr), np.asarray(hb.etype)
    dst = np.repeat(np.arange(hb.N), np.diff(rp))
    t_of = np.asarray(hb.graph_t)[np.asarray(hb.node_slot)]
    grp = (np.asarray(hb.node_slot) >= len(hb_s.graph_t)).astype(np.int64)      # slots are group-major
    assert np.all(grp[col] == grp[dst])
    for gsel, ref, shift in ((0, hb_s, 0), (1, hb_o, R)):
        m = grp[dst] == gsel
        got = np.stack((hb.node_ent[col[m]], hb.node_ent[dst[m]], t_of[dst[m]], et[m]), axis=1)
        rcol, rrp, ret = np.asarray(ref.col), np.asarray(ref.row_ptr), np.asarray(ref.etype)
        rdst = np.repeat(np.arange(ref.N), np.diff(rrp))
        rt = np.asarray(ref.graph_t)[np.asarray(ref.node_slot)]
        want = np.stack((ref.node_ent[rcol], ref.node_ent[rdst], rt[rdst], (ret + shift) % (2 * R)), axis=1)
        assert np.array_equal(got[np.lexsort(got.T[::-1])], want[np.lexsort(want.T[::-1])])


def test_merged_pass_equals_the_two_passes():
    import cpu_abi_emulation
    undo = cpu_abi_emulation.install()
    try:
        import model as M
        import parallel
        quads, num_ent, R = _data()
        gd = P.build_graph_dict(quads, R)
        hs, ho = P.HistoryIndex(quads, 's'), P.HistoryIndex(quads, 'o')
        torch.manual_seed(21)
        net = M.RENet(num_ent, 100, R, dropout=0.0, seq_len=10)
        gen = torch.Generator().manual_seed(2)
        net.global_emb = {int(t): torch.randn(1, 1, 100, generator=gen) * 0.1 for t in gd}
        net.eval()
        idx = np.random.RandomState(4).permutation(len(quads))[:160]
        b = quads[idx]
        flat = parallel.FlatGrads(net)
        ps = net.prepare(b, hs.take(idx), gd, subject=True)
        po = net.prepare(b, ho.take(idx), gd, subject=False)
        l1 = net.loss_prepared(ps) + net.loss_prepared(po)
        l1.backward()
        g1 = flat.flat.clone()
        flat.zero()
        pb = net.prepare_both(b, hs.take(idx), ho.take(idx), gd)
        assert pb.b == 2 * len(b) and pb.g.N == ps.g.N + po.g.N
        l2 = net.loss_prepared_both(pb)
        l2.backward()
        assert abs(float(l1) - float(l2)) <= 1e-6 * abs(float(l1))
        scale = float(g1.abs().max())
        assert float((g1 - flat.flat).abs().max()) <= 2e-6 * scale
        # ragged / degenerate batches: rows with empty histories on one or both sides mixed in, a single quadruple
        tmin = quads[:, 3].min()
        early = np.arange(len(quads))[quads[:, 3] <= tmin + 2][:24]
        late = np.arange(len(quads))[quads[:, 3] >= quads[:, 3].max() - 1][:9]
        for sel in (np.concatenate((early, late)), late[:1]):
            bb = quads[sel]
            flat.zero()
            la = net.loss_prepared(net.prepare(bb, hs.take(sel), gd, subject=True)) + \
                net.loss_prepared(net.prepare(bb, ho.take(sel), gd, subject=False))
            la.backward()
            ga = flat.flat.clone()
            flat.zero()
            pm = net.prepare_both(bb, hs.take(sel), ho.take(sel), gd)
            assert pm is not None and pm.b == 2 * len(bb)
            lm = net.loss_prepared_both(pm)
            lm.backward()
            assert abs(float(la) - float(lm)) <= 2e-6 * abs(float(la)), (len(sel), float(la), float(lm))
            assert float((ga - flat.flat).abs().max()) <= 4e-6 * float(ga.abs().max())
        # a direction without any history: callers fall back to the two separate passes
        first = np.arange(len(quads))[quads[:, 3] == quads[:, 3].min()][:8]
        assert net.prepare_both(quads[first], hs.take(first), ho.take(first), gd) is None
    finally:
        undo()


def test_fused_directions_of_forward_equal_the_two_calls_of_train_py():
    """RENet.fuse_directions (opt-in): train.py:136-138's  model(..., subject=True) + model(..., subject=False)  through
    the reference's list API run as ONE merged pass -- each call's VALUE is its own direction's loss, the sum carries the
    gradient of both; breaking the calling contract raises."""
    import cpu_abi_emulation
    undo = cpu_abi_emulation.install()
    try:
        import model as M
        import parallel
        quads, num_ent, R = _data()
        gd = P.build_graph_dict(quads, R)
        hs, ho = P.HistoryIndex(quads, 's'), P.HistoryIndex(quads, 'o')
        torch.manual_seed(21)
        net = M.RENet(num_ent, 100, R, dropout=0.0, seq_len=10)
        gen = torch.Generator().manual_seed(2)
        net.global_emb = {int(t): torch.randn(1, 1, 100, generator=gen) * 0.1 for t in gd}
        net.train()
        flat = parallel.FlatGrads(net)
        idx = np.random.RandomState(4).permutation(len(quads))[:160]
        bt = torch.from_numpy(quads[idx])
        (sh, sht), (oh, oht) = hs.to_lists(idx), ho.to_lists(idx)

        def two_calls():
            flat.zero()
            ls = net(bt, (sh, sht), (oh, oht), gd, subject=True)          # (fresh tuples per call, as train.py writes them)
            lo = net(bt, (sh, sht), (oh, oht), gd, subject=False)
            (ls + lo).backward()
            return float(ls), float(lo), flat.flat.clone()
        assert net.fuse_directions is False
        ls0, lo0, g0 = two_calls()
        net.fuse_directions = True
        ls1, lo1, g1 = two_calls()
        assert net._fused_pending is None
        assert abs(ls0 - ls1) <= 2e-6 * abs(ls0) and abs(lo0 - lo1) <= 2e-6 * abs(lo0)
        assert float((g0 - g1).abs().max()) <= 2e-6 * float(g0.abs().max())
        # eval mode / no_grad: the plain path (nothing pending afterwards)
        net.eval()
        with torch.no_grad():
            le = net(bt, (sh, sht), (oh, oht), gd, subject=True)
        assert net._fused_pending is None and abs(float(le) - ls0) <= 2e-6 * abs(ls0)
        net.train()
        # an object-direction call on its own is the plain pass
        assert abs(float(net(bt, (sh, sht), (oh, oht), gd, subject=False)) - lo0) <= 2e-6 * abs(lo0)
        # contract violations raise: two subject=True calls in a row; a subject=False call on other arguments
        net(bt, (sh, sht), (oh, oht), gd, subject=True)
        try:
            net(bt, (sh, sht), (oh, oht), gd, subject=True)
            raise AssertionError('expected RuntimeError')
        except RuntimeError as e:
            assert 'fuse_directions' in str(e)
        assert net._fused_pending is None
        net(bt, (sh, sht), (oh, oht), gd, subject=True)
        other = torch.from_numpy(quads[idx])
        try:
            net(other, (sh, sht), (oh, oht), gd, subject=False)
            raise AssertionError('expected RuntimeError')
        except RuntimeError as e:
            assert 'fuse_directions' in str(e)
    finally:
        undo()


def test_sequence_shards_of_one_batch_sum_to_the_batch():
    """SURVEY 8e option (i): every rank builds the SAME merged batch and keeps its share of the sequences
    (graph.shard_sequences); the ranks' losses and gradients SUM to those of the unsharded batch."""
    import cpu_abi_emulation
    undo = cpu_abi_emulation.install()
    try:
        import model as M
        import parallel
        quads, num_ent, R = _data()
        gd = P.build_graph_dict(quads, R)
        hs, ho = P.HistoryIndex(quads, 's'), P.HistoryIndex(quads, 'o')
        torch.manual_seed(5)
        net = M.RENet(num_ent, 100, R, dropout=0.0, seq_len=10)
        gen = torch.Generator().manual_seed(2)
        net.global_emb = {int(t): torch.randn(1, 1, 100, generator=gen) * 0.1 for t in gd}
        net.eval()
        idx = np.random.RandomState(9).permutation(len(quads))[:150]
        b = quads[idx]
        flat = parallel.FlatGrads(net)
        full = net.prepare_both(b, hs.take(idx), ho.take(idx), gd)
        lf = net.loss_prepared_both(full)
        lf.backward()
        gf = flat.flat.clone()
        for world in (2, 3, 8):
            flat.zero()
            total, rows, perms = 0.0, 0, []
            for rank in range(world):
                pr = net.prepare_both(b, hs.take(idx), ho.take(idx), gd, shard=(rank, world))
                assert pr.g.N == full.g.N and pr.g.E == full.g.E            # the same batch graph on every rank
                l = net.loss_prepared_both(pr)
                l.backward()                                                 # gradients accumulate = all-reduce SUM
                total += float(l)
                rows += pr.b
                perms.append(np.asarray(pr.perm))
            assert rows == full.b
            assert sorted(np.concatenate(perms).tolist()) == list(range(full.b))   # every sequence on exactly one rank
            assert abs(total - float(lf)) <= 2e-6 * abs(float(lf)), (world, total, float(lf))
            assert float((flat.flat - gf).abs().max()) <= 4e-6 * float(gf.abs().max()), world
    finally:
        undo()


def test_dual_head_function_equals_two_head_functions_over_emulated_kernels():
    """ops.DualHeadCEFn (both score heads + their weighted sum as one Function) against two ops.HeadCEFn calls combined
    by autograd, kernels emulated in torch-CPU: loss and every gradient, with and without pre-existing .grad buffers
    (the in-place accumulation targets of the training step)."""
    import cpu_abi_emulation
    import ops
    undo = cpu_abi_emulation.install()
    try:
        rng = np.random.RandomState(4)
        b, n_ent, n_rel, d = 48, 90, 14, 8
        ia = torch.from_numpy(rng.randint(0, n_ent, b).astype(np.int32))
        ic = torch.from_numpy(rng.randint(0, n_rel, b).astype(np.int32))
        t1 = torch.from_numpy(rng.randint(0, n_ent, b).astype(np.int32))
        t2 = torch.from_numpy(rng.randint(0, n_rel, b).astype(np.int32))

        def plan(idx):
            h = G.SegPlan.host(idx.numpy())
            p = G.SegPlan()
            p.order, p.seg_ptr, p.target = (torch.from_numpy(np.ascontiguousarray(a)) for a in
                                            (h.order, h.seg_ptr, h.target))
            p.num_segments = h.num_segments
            return p
        plan_a, plan_c = plan(ia), plan(ic)
        base = {n: torch.from_numpy(rng.randn(*shp).astype(np.float32) * 0.3) for n, shp in (
            ('ent', (n_ent, d)), ('rel', (n_rel, d)), ('h1', (b, d)), ('h2', (b, d)), ('w1', (n_ent, 3 * d)),
            ('b1', (n_ent,)), ('w2', (n_rel, 2 * d)), ('b2', (n_rel,)))}

        def run(dual, with_buffers):
            L = {n: v.clone().requires_grad_(True) for n, v in base.items()}
            if with_buffers:
                for n in ('ent', 'rel', 'w1', 'b1', 'w2', 'b2'):
                    L[n].grad = torch.full_like(L[n], 0.5)
            if dual:
                loss = ops.DualHeadCEFn.apply(L['ent'], ia, L['h1'], L['rel'], ic, L['w1'], L['b1'], t1, L['h2'], L['w2'],
                                              L['b2'], t2, plan_a, plan_c, 0.0, 0, 0, 2.0, 0.1)
            else:
                l1 = ops.HeadCEFn.apply(L['ent'], ia, L['h1'], L['rel'], ic, L['w1'], L['b1'], t1, plan_a, plan_c, 0.0,
                                        0, 2.0)
                l2 = ops.HeadCEFn.apply(L['ent'], ia, L['h2'], None, None, L['w2'], L['b2'], t2, plan_a, None, 0.0, 0,
                                        2.0)
                loss = l1 + 0.1 * l2
            loss.backward()
            return float(loss), {n: L[n].grad.clone() for n in L}
        for with_buffers in (False, True):
            l_ref, g_ref = run(False, with_buffers)
            l_new, g_new = run(True, with_buffers)
            assert abs(l_new - l_ref) <= 1e-6 * abs(l_ref)
            for n in g_ref:
                assert torch.allclose(g_new[n], g_ref[n], rtol=1e-5, atol=1e-7), n
    finally:
        undo()
