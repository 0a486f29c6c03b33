"""GPU test of the DEVICE batch-graph builder (csrc/builder.hip, gpu_builder.py): for batches of the ICEWS18- and
YAGO-shaped streams every array it produces is compared BIT FOR BIT with the host builder's (graph.build_batch_both =
the numpy specification + the native host passes, themselves pinned against the reference's DGL path in
tests/test_host_cpu.py), and a training step on a device-built batch equals the step on the host-built one."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a HIP device'
    import renet_hip
    renet_hip.lib()
    return torch.device('cuda:0')


def _setup(shape, seq_len, dev, num_t=None):
    import gpu_builder
    import preprocess as P
    import synth
    quads, ne, nr, _ = synth.make_stream(shape, seed=999, num_t=num_t)
    gd = P.build_graph_dict(quads, nr)
    hs, ho = P.HistoryIndex(quads, 's', seq_len), P.HistoryIndex(quads, 'o', seq_len)
    g = torch.Generator().manual_seed(3)
    glob = {int(t): torch.randn(1, 1, 8, generator=g) for t in gd}
    ds = gpu_builder.DeviceStore(quads, hs, ho, gd, glob, ne, nr, dev)
    return quads, ne, nr, gd, hs, ho, glob, ds


def _cmp(name, dev_t, host_a):
    a = dev_t.cpu().numpy() if dev_t is not None else np.zeros(0, np.int64)
    b = np.asarray(host_a)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    assert np.array_equal(a.astype(np.int64) if a.dtype != np.float32 else a, b.astype(np.int64) if b.dtype != np.float32 else b), \
        (name, np.nonzero(a.reshape(-1) != b.reshape(-1))[0][:10])


def _assert_same_batch(db, hb):
    for f in ('N', 'E', 'S', 'nnz', 'L', 'nA', 'E_out', 'n_chunks', 'n_chunks2', 'n_groups', 'n_groups_out', 'B'):
        assert int(getattr(db, f)) == int(getattr(hb, f)), (f, getattr(db, f), getattr(hb, f))
    for f in ('node_ent', 'node_slot', 'row_ptr', 'col', 'etype', 'norm', 'e_src', 'e_dst', 'chunk_ptr', 'chunk_type',
              'type_chunk_ptr', 'e_src2', 'e_dst2', 'chunk_ptr2', 'chunk_type2', 'type_chunk_ptr2', 'it_src', 'it_type',
              'grp_ptr', 'subj_row', 'row_ent', 'row_rel', 'glob_row', 's_sorted', 'r_sorted', 'rel_label', 'ent_label'):
        _cmp(f, getattr(db, f), getattr(hb, f))
    _cmp('heavy_rows', db.heavy_rows, hb.heavy_rows)
    _cmp('heavy_rows_out', db.heavy_rows_out, hb.heavy_rows_out)
    _cmp('step_off', db.step_off, hb.step_off)
    assert np.array_equal(db.host.step_off, hb.step_off) and np.array_equal(db.host.batch_sizes, hb.batch_sizes)
    assert np.array_equal(db.host.perm, hb.perm)
    for pn in ('plan_node_ent', 'plan_subj_row', 'plan_s', 'plan_r'):
        a, b = getattr(db, pn), getattr(hb, pn)
        assert a.num_segments == b.num_segments, pn
        for sub in ('order', 'seg_ptr', 'target'):
            _cmp(pn + '.' + sub, getattr(a, sub), getattr(b, sub))


@pytest.mark.parametrize('shape,seq_len,batch,num_t', [('ICEWS18', 10, 1024, None), ('YAGO', 15, 1024, None),
                                                       ('ICEWS18', 10, 96, 40), ('WIKI', 10, 777, 60)])
def test_device_built_batch_is_bit_identical_to_the_host_builder(dev, shape, seq_len, batch, num_t):
    import gpu_builder
    import graph as G
    quads, ne, nr, gd, hs, ho, glob, ds = _setup(shape, seq_len, dev, num_t)
    store = G.store_for(gd)
    gtimes = np.asarray(sorted(glob.keys()), dtype=np.int64)
    perm = np.random.RandomState(5).permutation(len(quads))
    for step in (0, 3, 11):
        idx = np.sort(perm[step * batch:(step + 1) * batch]) if step == 3 else perm[step * batch:(step + 1) * batch]
        if step == 11:
            idx = np.arange(batch)                          # the first quadruples of the stream: mostly EMPTY histories
        hb = G.build_batch_both(store, ne, nr, quads[idx, 0], quads[idx, 1], quads[idx, 2], hs.take(idx), ho.take(idx),
                                glob_index=lambda t: np.searchsorted(gtimes, t))
        db = None
        for attempt in range(6):                            # capacities grow on overflow
            db = gpu_builder.DeviceBatch(ds, idx, seq_len)
            if db.finalize():
                break
        assert db._final, 'device builder did not converge on capacities'
        _assert_same_batch(db, hb)


def test_training_step_on_a_device_built_batch_equals_the_host_built_one(dev):
    import model as M
    quads, ne, nr, gd, hs, ho, glob, ds = _setup('ICEWS18', 10, dev, 60)
    torch.manual_seed(1)
    net = M.RENet(ne, 200, nr, dropout=0.0, seq_len=10)
    net.global_emb = {t: torch.randn(1, 1, 200) * 0.1 for t in gd}
    import gpu_builder
    ds = gpu_builder.DeviceStore(quads, hs, ho, gd, net.global_emb, ne, nr, dev)
    net.to(dev).eval()
    idx = np.random.RandomState(2).permutation(len(quads))[:512]
    res = []
    for device_built in (False, True):
        for p in net.parameters():
            p.grad = None
        if device_built:
            prep = None
            while prep is None:
                prep = net.finish_prepare_device(net.prepare_both_device(idx, ds))
        else:
            prep = net.prepare_both(quads[idx], hs.take(idx), ho.take(idx), gd)
        loss = net.loss_prepared_both(prep)
        loss.backward()
        torch.cuda.synchronize()
        res.append((float(loss), {k: p.grad.clone() for k, p in net.named_parameters()}))
    assert res[0][0] == res[1][0]
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k


@pytest.mark.parametrize('shape,seq_len,batch,num_t', [('ICEWS18', 10, 1024, None), ('YAGO', 15, 512, 80)])
def test_list_api_batch_on_the_device_builder_is_bit_identical_to_the_host_builder(dev, shape, seq_len, batch, num_t):
    """gpu_builder.ListBatchStore: a batch that arrives as the reference's nested history lists (train.py:136-137), flattened
    and uploaded per call, through the same builder kernels -- every array equals the host builder's on the same lists."""
    import gpu_builder
    import graph as G
    quads, ne, nr, gd, hs, ho, glob, _ = _setup(shape, seq_len, dev, num_t)
    store = G.store_for(gd)
    gtimes = np.asarray(sorted(glob.keys()), dtype=np.int64)
    base = gpu_builder.graph_store_for(gd, glob, ne, nr, dev)
    assert gpu_builder.graph_store_for(gd, glob, ne, nr, dev) is base            # cached
    perm = np.random.RandomState(6).permutation(len(quads))
    for step in (0, 2):
        idx = perm[step * batch:(step + 1) * batch] if step == 0 else np.arange(40, 40 + batch)
        (sh, sht), (oh, oht) = hs.to_lists(idx), ho.to_lists(idx)
        fs, fo = G.FlatHistory.from_lists(sh, sht), G.FlatHistory.from_lists(oh, oht)
        hb = G.build_batch_both(store, ne, nr, quads[idx, 0], quads[idx, 1], quads[idx, 2], fs, fo,
                                glob_index=lambda t: np.searchsorted(gtimes, t))
        db = None
        for attempt in range(6):
            db = gpu_builder.DeviceBatch(gpu_builder.ListBatchStore(base, quads[idx], fs, fo), np.arange(len(idx)), seq_len)
            if db.finalize():
                break
        assert db._final
        _assert_same_batch(db, hb)


def test_fused_forward_on_the_device_builder_equals_the_host_builder(dev):
    """RENet.forward with fuse_directions through the list API: device-built merged batch (the default on a GPU) vs the host
    builder -- identical losses and gradients, bit for bit (same arrays, same kernels)."""
    import model as M
    quads, ne, nr, gd, hs, ho, glob, _ = _setup('ICEWS18', 10, dev, 60)
    torch.manual_seed(1)
    net = M.RENet(ne, 200, nr, dropout=0.0, seq_len=10)
    net.global_emb = {t: torch.randn(1, 1, 200) * 0.1 for t in gd}
    net.to(dev).train()
    net.fuse_directions = True
    idx = np.random.RandomState(2).permutation(len(quads))[:512]
    bt = torch.from_numpy(quads[idx]).to(dev)
    (sh, sht), (oh, oht) = hs.to_lists(idx), ho.to_lists(idx)
    res = []
    for on_device in (False, True):
        net.device_builder_lists = on_device
        net.zero_grad()
        ls = net(bt, (sh, sht), (oh, oht), gd, subject=True)
        lo = net(bt, (sh, sht), (oh, oht), gd, subject=False)
        (ls + lo).backward()
        torch.cuda.synchronize()
        res.append((float(ls), float(lo), {k: p.grad.clone() for k, p in net.named_parameters()}))
    assert res[0][0] == res[1][0] and res[0][1] == res[1][1], (res[0][:2], res[1][:2])
    for k in res[0][2]:
        assert torch.equal(res[0][2][k], res[1][2][k]), k
