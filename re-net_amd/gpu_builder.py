"""Device-side batch-graph builder for the merged training batch (csrc/builder.hip, renet_build_batch_both).

The host builder (graph.build_batch_both: numpy front + the native passes of csrc/host_builder.cpp) costs ~13 ms per
merged batch against a ~3.6 ms device step.  Here the dataset is resident in HBM -- the quadruples, the per-role
history index of preprocess.HistoryIndex and the per-timestamp fact lists of graph.GraphStore (`DeviceStore`) -- and a
batch is built from its B quadruple indices by kernels: 4 KB up, ~100 bytes of counts back, no synchronisation in
between.  `DeviceBatch` then exposes the same attributes as graph.DeviceGraph (the arrays are bit-identical to the host
builder's: tests/test_gpu_builder.py), so model.RENet.loss_prepared_both runs on it unchanged.
Replaces, for that case, utils.py:209-244 + 115-131 + dgl.batch of the reference."""
import ctypes

import numpy as np
import torch

import graph as G
import renet_hip as K

_P = ctypes.c_void_p
NCOUNTS = 64
(C_NNZ, C_S, C_L, C_TB, C_FACTS, C_N, C_NA, C_E2, C_E, C_NHEAVY, C_NHEAVY_OUT, C_NCHUNKS, C_NCHUNKS2, C_NITEMS,
 C_NGROUPS, C_NGROUPS_OUT, C_NSEG0, C_NSEG1, C_NSEG2, C_NSEG3, C_ERR, C_EOUT) = range(22)
C_STEP_OFF = 24
MAXL = 32


class _StoreDev(ctypes.Structure):
    _fields_ = [('q_s', _P), ('q_r', _P), ('q_o', _P), ('h_first', _P * 2), ('h_count', _P * 2), ('snap_t', _P * 2),
                ('snap_ptr', _P * 2), ('nbr_o', _P * 2), ('times', _P), ('trip_ptr', _P), ('trip_s', _P), ('trip_r', _P),
                ('trip_o', _P), ('glob_times', _P), ('T', ctypes.c_int), ('n_glob', ctypes.c_int),
                ('n_facts', ctypes.c_int), ('num_ent', ctypes.c_int), ('num_rels', ctypes.c_int)]


class _BatchOut(ctypes.Structure):
    _fields_ = [(n, _P) for n in ('node_ent', 'node_slot', 'row_ptr', 'col', 'etype', 'norm', 'heavy_rows', 'e_src', 'e_dst',
                                  'chunk_ptr', 'chunk_type', 'type_chunk_ptr', 'e_src2', 'e_dst2', 'chunk_ptr2',
                                  'chunk_type2', 'type_chunk_ptr2', 'it_src', 'it_type', 'grp_ptr', 'subj_row', 'row_seq',
                                  'row_ent', 'row_rel', 'glob_row', 's_sorted', 'r_sorted', 'rel_label', 'ent_label', 'perm',
                                  'step_off')] + \
               [('plan_order', _P * 4), ('plan_seg', _P * 4), ('plan_target', _P * 4), ('counts', _P),
                ('cap_nodes', ctypes.c_int), ('cap_edges', ctypes.c_int)]


def _bind():
    return K.lib()          # renet_build_batch_workspace / renet_build_batch_both are bound with the rest of the C ABI


def _i32(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)


class DeviceStore(object):
    """The dataset, resident on the device: quadruples, both history indices, the graph store, the sorted timestamps of
    the global-embedding table."""

    def __init__(self, quads, hist_s, hist_o, graph_dict, global_emb, num_ent, num_rels, device):
        quads = np.asarray(quads, dtype=np.int64)
        store = G.store_for(graph_dict)
        times = np.asarray(store.times, dtype=np.int64)
        if len(times) > 1 and np.any(np.diff(times) <= 0):
            raise ValueError('graph_dict must be in ascending time order')
        for arr in (quads, store.trip_s, store.trip_o, times):
            if len(arr) and (arr.max() >= 2 ** 31 or arr.min() < 0):
                raise ValueError('ids / timestamps must fit int32')
        self.device = device
        self.num_ent, self.num_rels = int(num_ent), int(num_rels)
        self.n_quads = len(quads)
        self.t = dict(q_s=_i32(quads[:, 0], device), q_r=_i32(quads[:, 1], device), q_o=_i32(quads[:, 2], device),
                      times=_i32(times, device), trip_ptr=_i32(store.trip_ptr, device), trip_s=_i32(store.trip_s, device),
                      trip_r=_i32(store.trip_r, device), trip_o=_i32(store.trip_o, device),
                      glob_times=_i32(sorted(int(t) for t in global_emb.keys()), device))
        for r, h in enumerate((hist_s, hist_o)):
            self.t['h_first%d' % r] = _i32(h.first, device)
            self.t['h_count%d' % r] = _i32(h.count, device)
            self.t['snap_t%d' % r] = _i32(h.snap_t, device)
            self.t['snap_ptr%d' % r] = _i32(h.snap_ptr, device)
            self.t['nbr_o%d' % r] = _i32(h.nbr_o, device)
        self.history_len = int(hist_s.history_len)
        self.n_facts = int(len(store.trip_s))
        sd = _StoreDev()
        for n in ('q_s', 'q_r', 'q_o', 'times', 'trip_ptr', 'trip_s', 'trip_r', 'trip_o', 'glob_times'):
            setattr(sd, n, self.t[n].data_ptr())
        for n in ('h_first', 'h_count', 'snap_t', 'snap_ptr', 'nbr_o'):
            setattr(sd, n, (_P * 2)(self.t[n + '0'].data_ptr(), self.t[n + '1'].data_ptr()))
        sd.T, sd.n_glob, sd.n_facts = len(times), int(self.t['glob_times'].numel()), self.n_facts
        sd.num_ent, sd.num_rels = self.num_ent, self.num_rels
        self.c = sd
        # capacities: grown on demand (an overflow is reported in the counts and the batch rebuilt)
        self.cap_nodes, self.cap_edges = 1 << 17, 1 << 19
        # pinned 256-byte count buffers: a free-list, not a ring -- every DeviceBatch OWNS its buffer from the async
        # D2H copy until finalize() has read it (with a shared ring of 8, the ninth pending batch overwrote
        # the counts of the first)
        self._pinned_free = []

    def _take_pinned(self):
        if self._pinned_free:
            return self._pinned_free.pop()
        return torch.empty(NCOUNTS, dtype=torch.int32, pin_memory=True)

    def _give_pinned(self, buf):
        if buf is not None and len(self._pinned_free) < 64:
            self._pinned_free.append(buf)


class GraphDeviceStore(DeviceStore):
    """The part of a DeviceStore that a batch does NOT bring itself: the per-timestamp fact lists of the graph store and the
    timestamps of the global-embedding table.  Used by the reference's LIST API (RENet.forward with nested history lists,
    train.py:136-137): there the quadruples and their histories arrive with every call (ListBatchStore uploads them, ~160 KB),
    so no dataset has to be registered with the model.  Cached per (graph store, global_emb size) in graph_store_for()."""

    def __init__(self, graph_dict, global_emb, num_ent, num_rels, device):      # (no DeviceStore.__init__: no dataset here)
        store = G.store_for(graph_dict)
        times = np.asarray(store.times, dtype=np.int64)
        if len(times) > 1 and np.any(np.diff(times) <= 0):
            raise ValueError('graph_dict must be in ascending time order')
        for arr in (store.trip_s, store.trip_o, times):
            if len(arr) and (arr.max() >= 2 ** 31 or arr.min() < 0):
                raise ValueError('ids / timestamps must fit int32')
        self.device = device
        self.num_ent, self.num_rels = int(num_ent), int(num_rels)
        self.n_quads = 0
        self.t = dict(times=_i32(times, device), trip_ptr=_i32(store.trip_ptr, device), trip_s=_i32(store.trip_s, device),
                      trip_r=_i32(store.trip_r, device), trip_o=_i32(store.trip_o, device),
                      glob_times=_i32(sorted(int(t) for t in global_emb.keys()), device))
        self.n_facts = int(len(store.trip_s))
        self.T, self.n_glob = len(times), int(self.t['glob_times'].numel())
        self.cap_nodes, self.cap_edges = 1 << 17, 1 << 19
        self._pinned_free = []
        self.c = None                         # (a ListBatchStore fills a struct per batch)


_graph_stores = {}


def graph_store_for(graph_dict, global_emb, num_ent, num_rels, device):
    """The resident GraphDeviceStore of (graph_dict, global_emb) on `device`; rebuilt when the graph store object changed
    (graph.store_for re-creates it when the dict gained timestamps) or the global-embedding table's TIMESTAMPS changed (the
    store holds their sorted list: the same count with different keys must not reuse it).  Entries die with their
    graph_dict (weak reference), at most 8 stay resident."""
    import weakref
    store = G.store_for(graph_dict)
    key = (id(graph_dict), str(device))
    stamp = hash(tuple(sorted(int(t) for t in global_emb.keys())))
    ent = _graph_stores.get(key)
    if ent is None or ent[0] is not store or ent[1] != stamp:
        if len(_graph_stores) > 8:
            _graph_stores.clear()
        ent = (store, stamp, GraphDeviceStore(graph_dict, global_emb, num_ent, num_rels, device))
        _graph_stores[key] = ent
        try:
            weakref.finalize(graph_dict, _graph_stores.pop, key, None)
        except TypeError:                    # (a plain dict cannot be weakly referenced: the size bound above applies)
            pass
    return ent[2]


class ListBatchStore(object):
    """What DeviceBatch needs of a store, for ONE batch that arrives through the reference's list API: the batch's own
    quadruples (s, r, o) and its two FlatHistory objects (graph.FlatHistory.from_lists of the nested lists) uploaded in ONE
    int32 copy, laid out like a DeviceStore's history index (first / count per sequence, snapshot timestamps, snapshot
    pointers, neighbours) on top of the resident GraphDeviceStore.  The batch's quadruple indices are 0..B-1."""

    def __init__(self, base, trip, fs, fo, stream=None):
        self.base = base
        self.device, self.num_ent, self.num_rels = base.device, base.num_ent, base.num_rels
        trip = np.asarray(trip, dtype=np.int64)
        B = len(trip)
        if len(fs) != B or len(fo) != B:
            raise ValueError('histories and quadruples differ in length')
        parts = [trip[:, 0], trip[:, 1], trip[:, 2]]
        for fh in (fs, fo):
            parts += [fh.seq_ptr[:-1], np.diff(fh.seq_ptr), fh.step_t, fh.nbr_ptr, fh.nbr_o]
        for a in parts:
            if len(a) and (int(a.max()) >= 2 ** 31 or int(a.min()) < 0):
                raise ValueError('ids / timestamps must fit int32')
        offs, tot = [], 0
        for a in parts:
            offs.append(tot)
            tot += (len(a) + 15) & ~15                       # 64-byte aligned slices
        flat = np.zeros(tot + 16, dtype=np.int32)
        for a, o in zip(parts, offs):
            flat[o:o + len(a)] = a
        st = stream if stream is not None else torch.cuda.current_stream()
        with torch.cuda.stream(st):
            self._buf = torch.from_numpy(flat).to(self.device, non_blocking=True)
        ptr = [self._buf.data_ptr() + 4 * o for o in offs]
        sd = _StoreDev()
        sd.q_s, sd.q_r, sd.q_o = ptr[0], ptr[1], ptr[2]
        for k, n in enumerate(('h_first', 'h_count', 'snap_t', 'snap_ptr', 'nbr_o')):
            setattr(sd, n, (_P * 2)(ptr[3 + k], ptr[8 + k]))
        for n in ('times', 'trip_ptr', 'trip_s', 'trip_r', 'trip_o', 'glob_times'):
            setattr(sd, n, base.t[n].data_ptr())
        sd.T, sd.n_glob, sd.n_facts = base.T, base.n_glob, base.n_facts
        sd.num_ent, sd.num_rels = self.num_ent, self.num_rels
        self.c = sd
        self.n_quads = B

    # capacities and the pinned count buffers live in the resident store (they outlast the batch)
    cap_nodes = property(lambda self: self.base.cap_nodes, lambda self, v: setattr(self.base, 'cap_nodes', v))
    cap_edges = property(lambda self: self.base.cap_edges, lambda self, v: setattr(self.base, 'cap_edges', v))

    def _take_pinned(self):
        return self.base._take_pinned()

    def _give_pinned(self, buf):
        self.base._give_pinned(buf)


class _Host(object):
    """Host-side view of a DeviceBatch (what graph._HostView offers for a host-built batch)."""

    def __init__(self, batch, c):
        self._batch = batch
        self.N, self.E, self.S, self.nnz, self.L, self.nA, self.B = batch.N, batch.E, batch.S, batch.nnz, batch.L, batch.nA, batch.B
        self.step_off = c[C_STEP_OFF:C_STEP_OFF + batch.L + 1].astype(np.int32)
        self.batch_sizes = np.diff(self.step_off).astype(np.int64)
        self._perm = None

    @property
    def perm(self):
        if self._perm is None:              # sorted position -> sequence of the merged batch (one small D2H, on demand)
            self._perm = self._batch._v['perm'].cpu().numpy().astype(np.int64)
        return self._perm


class DeviceBatch(object):
    """A merged batch built on the device.  After finalize() it carries the attributes of graph.DeviceGraph."""

    def __init__(self, store, idx, seq_len, stream=None):
        L = _bind()
        self.store, dev = store, store.device
        self.B = int(len(idx))
        B2 = 2 * self.B
        cn, ce = store.cap_nodes, store.cap_edges
        cs = B2 * MAXL
        T2 = 2 * store.num_rels
        cc = ce // G.CHUNK + T2 + 2
        sizes = dict(node_ent=cn, node_slot=cn, row_ptr=cn + 2, col=ce, etype=ce, heavy_rows=cn, e_src=ce, e_dst=ce,
                     chunk_ptr=cc + 1, chunk_type=cc, type_chunk_ptr=T2 + 1, e_src2=ce, e_dst2=ce, chunk_ptr2=cc + 1,
                     chunk_type2=cc, type_chunk_ptr2=T2 + 1, it_src=ce + cn, it_type=ce + cn, grp_ptr=cn + 2,
                     subj_row=cs, row_seq=cs, row_ent=cs, row_rel=cs, glob_row=cs, s_sorted=B2, r_sorted=B2,
                     rel_label=B2, ent_label=B2, perm=B2, step_off=MAXL + 1, counts=NCOUNTS,
                     plan_order0=cn, plan_seg0=cn + 1, plan_target0=cn, plan_order1=cs, plan_seg1=cs + 1, plan_target1=cs,
                     plan_order2=B2, plan_seg2=B2 + 1, plan_target2=B2, plan_order3=B2, plan_seg3=B2 + 1, plan_target3=B2)
        offs, tot = {}, 0
        for n, m in sizes.items():
            offs[n] = tot
            tot += (m + 63) & ~63
        st = stream if stream is not None else torch.cuda.current_stream()
        with torch.cuda.stream(st):              # the buffers belong to the stream the builder runs on
            self._buf = torch.empty(tot, device=dev, dtype=torch.int32)
            self._norm = torch.empty(cn, device=dev, dtype=torch.float32)
        self._v = {n: self._buf[o:o + sizes[n]] for n, o in offs.items()}
        out = _BatchOut()
        for n, _ in _BatchOut._fields_:
            if n in self._v:
                setattr(out, n, self._v[n].data_ptr())
        out.norm = self._norm.data_ptr()
        for f in ('plan_order', 'plan_seg', 'plan_target'):
            setattr(out, f, (_P * 4)(*[self._v['%s%d' % (f, k)].data_ptr() for k in range(4)]))
        out.counts = self._v['counts'].data_ptr()
        out.cap_nodes, out.cap_edges = cn, ce
        self._caps = (cn, ce)
        nbytes = L.renet_build_batch_workspace(ctypes.addressof(store.c), self.B, cn, ce)
        with torch.cuda.stream(st):
            ws = torch.empty(nbytes // 4 + 64, device=dev, dtype=torch.int32)
            self._idx = torch.from_numpy(np.ascontiguousarray(idx, dtype=np.int32)).to(dev, non_blocking=True)
        self._out = out                      # (kept alive: the call reads the struct through its address)
        rc = L.renet_build_batch_both(ctypes.addressof(store.c), self._idx.data_ptr(), self.B, int(seq_len), G.HEAVY,
                                      G.GROUP_ITEMS, G.CHUNK, ctypes.addressof(out), ws.data_ptr(), nbytes, st.cuda_stream)
        if rc != 0:
            raise K.RenetHipError('renet_build_batch_both failed with code %d' % rc)
        self._ws = ws                       # stays alive until the kernels have run (freed in finalize)
        self._counts_host = store._take_pinned()     # owned by this batch until finalize() / release
        self._stream = st
        with torch.cuda.stream(st):
            self._counts_host.copy_(self._v['counts'], non_blocking=True)
            self._done = torch.cuda.Event()
            self._done.record(st)
        self._final = False

    def finalize(self):
        """Waits for the counts, slices the outputs to their sizes, composes the table-addressed item arrays.
        Returns False if a capacity was exceeded (the store's capacities are then doubled: rebuild)."""
        if self._final:
            return True
        self._done.synchronize()
        c = self._counts_host.numpy().astype(np.int64)      # (a copy: the pinned buffer goes back to the free-list)
        self.store._give_pinned(self._counts_host)
        self._counts_host = None
        self._ws = None
        cur = torch.cuda.current_stream()
        if cur != self._stream:                  # built on a side stream, consumed on this one
            self._buf.record_stream(cur)
            self._norm.record_stream(cur)
        if c[C_ERR] != 0:
            if c[C_ERR] & 4:
                self.store.cap_nodes *= 2
            if c[C_ERR] & 8:
                self.store.cap_edges *= 2
            if c[C_ERR] & 3:
                raise KeyError('a history timestamp is missing from graph_dict / global_emb')
            return False
        v = self._v
        N, nA, E, S = int(c[C_N]), int(c[C_NA]), int(c[C_E]), int(c[C_S])
        self.N, self.nA, self.E, self.S, self.nnz, self.L = N, nA, E, S, int(c[C_NNZ]), int(c[C_L])
        self.B = 2 * self.B                      # sequences of the merged batch (graph.HostBatch.B)
        self.num_types = 2 * self.store.num_rels
        self.heavy_thresh = G.HEAVY
        self.node_ent, self.node_slot = v['node_ent'][:N], v['node_slot'][:N]
        self.row_ptr, self.col, self.etype = v['row_ptr'][:N + 1], v['col'][:E], v['etype'][:E]
        self.norm = self._norm[:N]
        nh, nho = int(c[C_NHEAVY]), int(c[C_NHEAVY_OUT])
        self.heavy_rows = v['heavy_rows'][:nh] if nh else None
        self.heavy_rows_out = v['heavy_rows'][:nho] if nho else None
        self.e_src, self.e_dst = v['e_src'][:E], v['e_dst'][:E]
        self.n_chunks, self.n_chunks2 = int(c[C_NCHUNKS]), int(c[C_NCHUNKS2])
        self.chunk_ptr, self.chunk_type = v['chunk_ptr'][:self.n_chunks + 1], v['chunk_type'][:self.n_chunks]
        self.type_chunk_ptr, self.type_chunk_ptr2 = v['type_chunk_ptr'], v['type_chunk_ptr2']
        self.chunk_ptr2, self.chunk_type2 = v['chunk_ptr2'][:self.n_chunks2 + 1], v['chunk_type2'][:self.n_chunks2]
        self.E_out = int(c[C_EOUT])
        self.e_src2, self.e_dst2 = v['e_src2'][:self.E_out], v['e_dst2'][:self.E_out]
        ni = int(c[C_NITEMS])
        self.it_src, self.it_type = v['it_src'][:ni], v['it_type'][:ni]
        self.n_groups, self.n_groups_out = int(c[C_NGROUPS]), int(c[C_NGROUPS_OUT])
        self.grp_ptr = v['grp_ptr'][:self.n_groups + 1]
        for n in ('subj_row', 'row_seq', 'row_ent', 'row_rel', 'glob_row'):
            setattr(self, n, v[n][:S])
        for n in ('s_sorted', 'r_sorted', 'rel_label', 'ent_label'):
            setattr(self, n, v[n])
        self.step_off = v['step_off'][:self.L + 1]
        for k, name in enumerate(('plan_node_ent', 'plan_subj_row', 'plan_s', 'plan_r')):
            p = G.SegPlan()
            u = int(c[C_NSEG0 + k])
            n_rows = (N, S, self.B, self.B)[k]
            p.order, p.seg_ptr, p.target = v['plan_order%d' % k][:n_rows], v['plan_seg%d' % k][:u + 1], v['plan_target%d' % k][:u]
            p.num_segments = u
            setattr(self, name, p)
        self.ndata = {}
        self._table_items = None
        # the few host-side values the model still needs (packing of the GRU launches, reporting)
        self.host = _Host(self, c)
        self._final = True
        return True

    def __del__(self):
        # a batch dropped before finalize(): its copy may still be in flight -- wait for it before the buffer is reused
        try:
            if getattr(self, '_counts_host', None) is not None:
                self._done.synchronize()
                self.store._give_pinned(self._counts_host)
                self._counts_host = None
        except Exception:       # interpreter shutdown
            pass

    def table_items(self):
        if self._table_items is None:
            self._table_items = K.compose_table_items(self)
        return self._table_items
