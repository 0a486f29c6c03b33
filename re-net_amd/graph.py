"""DGL-free graph format and the (host-side, vectorised) batch-graph builder.

Replaces, for the hot path, the reference's use of DGL graph objects:
  * `TimeGraph`          <- the per-timestamp DGLGraph of utils.py:68-87 (get_big_graph)
  * `GraphStore`         <- graph_dict flattened into timestamp-indexed triple arrays
  * `build_batch(...)`   <- utils.py:209-283 get_sorted_s_r_embed_rgcn / get_s_r_embed_rgcn +
                            utils.py:115-131 make_subgraph + dgl.batch, i.e. ~T_b DGL subgraph
                            calls and Python dict/set loops per pass, as a handful of numpy
                            sort/searchsorted passes that emit directly what the HIP kernels
                            consume: CSR-by-destination with relation-sorted rows, a relation-bucketed
                            edge list for dW, the packed (time-major) sequence layout and the sorted
                            plans for the deterministic segmented scatter-adds of the backward pass.

Semantics kept from the reference: the subgraph at time t is induced on the union,
over ALL sequences of the batch, of {subject} U {objects in the history step at t}; `norm` is
1/in-degree of that induced subgraph.  Node order inside a member graph is by entity id (the
reference's is set-iteration order; results do not depend on it).
"""
import contextlib
from itertools import chain as _chain

try:                                   # host-side list flattener (re-net_amd/build.py builds it next to this file)
    import _renet_listwalk as _listwalk
except ImportError:
    _listwalk = None

import numpy as np
import torch

import ctypes

CHUNK = 64          # edges per dW work item (one wave each)
NATIVE = True       # use csrc/host_builder.cpp for the heavy passes (the numpy code below is their executable
                    # specification; tests/test_host_cpu.py compares the two bit for bit)


def _native():
    """The C library's host-builder entry points, or None when disabled (no GPU is needed for them)."""
    if not NATIVE:
        return None
    import renet_hip
    return renet_hip.lib()


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)
import os as _os
# in-degree above which a row is reduced by a whole workgroup (hub rows), and the item budget of one wave's
# row group in the gather kernel's item stream (plan_gather_items); swept in tools/gather_bench.py
HEAVY = int(_os.environ.get('RENET_GATHER_HEAVY', '8'))
GROUP_ITEMS = int(_os.environ.get('RENET_GATHER_GROUP', '12'))


def plan_gather_items(row_ptr, col, etype, n_out, heavy, budget):
    """Item stream of the gather-SpMM kernels (renet_rgcn_gather_items) -- numpy specification of
    csrc/host_builder.cpp:renet_host_gather_items.

    Every LIGHT row v (in-degree <= heavy) contributes its in-edges, in CSR order, as items (col[e], etype[e])
    followed by one FLUSH item (v, -1); hub rows contribute nothing (a workgroup each reduces them).  The
    stream is cut into groups, one per wave: a new group starts at the first light row, whenever a row's first
    item falls into a new `budget`-sized window of the stream, and at the first light row >= n_out (so the
    groups of rows < n_out are a prefix: the pruned last layer launches only those).  A group therefore holds
    fewer than budget + heavy + 1 <= 64 items.
    Returns (it_src[I], it_type[I], grp_ptr[G+1], n_groups_out) as int32 arrays / int."""
    if budget + heavy + 1 > 64 or budget < 1 or heavy < 0:
        raise ValueError('gather plan needs budget + heavy + 1 <= 64')
    row_ptr = np.asarray(row_ptr, dtype=np.int64)
    n = len(row_ptr) - 1
    deg = np.diff(row_ptr)
    light = np.nonzero(deg <= heavy)[0]
    cnt = deg[light] + 1
    start = np.concatenate(([0], np.cumsum(cnt)))                  # item offset of every light row (+ total)
    total = int(start[-1])
    it_src = np.empty(total, np.int32)
    it_type = np.empty(total, np.int32)
    if len(light):
        e_idx = ragged_arange(row_ptr[light], deg[light])           # the light rows' edges, CSR order
        e_pos = ragged_arange(start[:-1], deg[light])
        it_src[e_pos] = np.asarray(col)[e_idx]
        it_type[e_pos] = np.asarray(etype)[e_idx]
        fpos = start[:-1] + deg[light]
        it_src[fpos] = light
        it_type[fpos] = -1
        key = start[:-1] // budget
        side = light >= n_out
        first = np.ones(len(light), dtype=bool)
        first[1:] = (key[1:] != key[:-1]) | (side[1:] != side[:-1])
        grp_ptr = np.concatenate((start[:-1][first], [total])).astype(np.int32)
        n_groups_out = int(np.count_nonzero(first & ~side))
    else:
        grp_ptr = np.zeros(1, np.int32)
        n_groups_out = 0
    return it_src, it_type, grp_ptr, n_groups_out


class TimeGraph(object):
    """Per-timestamp multigraph.  `ent` sorted unique entity ids; facts as local triples (ls, r, lo);
    both directions of every fact are implied: ls->lo with (type_s, type_o) = (r, r+R) and
    lo->ls with (r+R, r)  (utils.py:74-76)."""
    __slots__ = ('ent', 'ls', 'r', 'lo', 'num_rels', '_ids')

    def __init__(self, ent, ls, r, lo, num_rels):
        self.ent = np.ascontiguousarray(ent, dtype=np.int64)
        self.ls = np.ascontiguousarray(ls, dtype=np.int64)
        self.r = np.ascontiguousarray(r, dtype=np.int64)
        self.lo = np.ascontiguousarray(lo, dtype=np.int64)
        self.num_rels = int(num_rels)
        self._ids = None

    @classmethod
    def from_triples(cls, triples, num_rels):
        triples = np.asarray(triples, dtype=np.int64).reshape(-1, 3)
        ent, inv = np.unique(np.stack((triples[:, 0], triples[:, 2])), return_inverse=True)
        ls, lo = inv.reshape(2, -1)
        return cls(ent, ls, triples[:, 1], lo, num_rels)

    # -- small DGL-flavoured surface the reference's callers use on graph_dict entries ----------
    def number_of_nodes(self):
        return int(self.ent.shape[0])

    def number_of_edges(self):
        return 2 * int(self.ls.shape[0])

    @property
    def ids(self):
        """entity id -> local node (the `g.ids` dict of utils.py:82-86), built on demand."""
        if self._ids is None:
            self._ids = dict(zip(self.ent.tolist(), range(len(self.ent))))
        return self._ids

    def edges(self, reverse=False):
        """(src, dst, etype) of the directed edges; etype = type_o if reverse else type_s."""
        src = np.concatenate((self.ls, self.lo))
        dst = np.concatenate((self.lo, self.ls))
        if reverse:
            et = np.concatenate((self.r + self.num_rels, self.r))
        else:
            et = np.concatenate((self.r, self.r + self.num_rels))
        return src, dst, et

    def global_triples(self):
        return self.ent[self.ls], self.r, self.ent[self.lo]

    def __getstate__(self):
        return (self.ent, self.ls, self.r, self.lo, self.num_rels)

    def __setstate__(self, st):
        self.ent, self.ls, self.r, self.lo, self.num_rels = st
        self._ids = None


class GraphStore(object):
    """All per-timestamp graphs of a graph_dict as flat, timestamp-indexed triple arrays."""

    def __init__(self, graph_dict):
        self.times = np.asarray(list(graph_dict.keys()), dtype=np.int64)
        self._sig = self.signature(graph_dict)
        gs = [graph_dict[t] for t in graph_dict]
        # strong references: the signature compares object identities, and a freed TimeGraph's address can be
        # handed to a NEW graph (same-shaped dicts rebuilt in a loop do exactly that) -- while the store lives,
        # the ids it was built from cannot be reused
        self._graphs = gs
        self._keys = tuple(graph_dict.keys())
        cnt = np.asarray([len(g.ls) for g in gs], dtype=np.int64)
        self.trip_ptr = np.concatenate(([0], np.cumsum(cnt)))
        if gs:
            self.trip_s = np.concatenate([g.ent[g.ls] for g in gs])
            self.trip_r = np.concatenate([g.r for g in gs])
            self.trip_o = np.concatenate([g.ent[g.lo] for g in gs])
        else:
            self.trip_s = self.trip_r = self.trip_o = np.zeros(0, np.int64)
        self.node_cnt = np.asarray([g.number_of_nodes() for g in gs], dtype=np.int64)
        # the native passes index scratch tables with these ids unchecked: validate once, here
        self.max_ent = int(max(self.trip_s.max(), self.trip_o.max())) if len(self.trip_s) else -1
        self.max_rel = int(self.trip_r.max()) if len(self.trip_r) else -1
        if len(self.trip_s) and (min(self.trip_s.min(), self.trip_o.min()) < 0 or self.trip_r.min() < 0):
            raise ValueError('negative entity / relation id in graph_dict')
        order = np.argsort(self.times, kind='stable')
        self._sorted_times = self.times[order]
        self._sorted_pos = order

    def subject_index(self):
        """(by_subj, subj_sorted): fact indices sorted stably by (timestamp, subject) and their subjects -- built
        on first use (batched inference walks only the facts of a member graph's own nodes)."""
        if getattr(self, '_by_subj', None) is None:
            t_of = np.repeat(np.arange(len(self.trip_ptr) - 1, dtype=np.int64), np.diff(self.trip_ptr))
            self._by_subj = np.ascontiguousarray(np.lexsort((self.trip_s, t_of)), dtype=np.int64)
            self._subj_sorted = np.ascontiguousarray(self.trip_s[self._by_subj], dtype=np.int64)
        return self._by_subj, self._subj_sorted

    @staticmethod
    def signature(graph_dict):
        return (id(graph_dict), len(graph_dict), tuple(id(g) for g in graph_dict.values()))

    def matches(self, graph_dict):
        return self._sig == self.signature(graph_dict) and self._keys == tuple(graph_dict.keys())

    def index_of(self, t):
        p = np.searchsorted(self._sorted_times, t)
        if np.any(p >= len(self._sorted_times)) or np.any(self._sorted_times[np.minimum(p, len(self._sorted_times) - 1)] != t):
            raise KeyError('timestamp not in graph_dict')
        return self._sorted_pos[p]


_store_cache = {}


def store_for(graph_dict):
    key = id(graph_dict)
    st = _store_cache.get(key)
    if st is None or not st.matches(graph_dict):
        st = GraphStore(graph_dict)
        _store_cache.clear()          # one live graph_dict at a time is the norm
        _store_cache[key] = st
    return st


import threading

_scratch = threading.local()


def _lookup_table(n):
    """Reusable per-thread int32 scratch table pre-filled with -1 (callers restore the entries they touch)."""
    t = getattr(_scratch, 'table', None)
    if t is None or len(t) < n:
        t = np.full(max(n, 1 << 20), -1, dtype=np.int32)
        _scratch.table = t
    return t


def stable_argsort(keys, bound):
    """Stable argsort of non-negative integer keys < bound.  numpy's stable sort is an O(n) radix sort only
    for 16-bit keys (for int64 it is a mergesort ~6x slower at the sizes of a batch), so sort by 16-bit digits
    least-significant first."""
    keys = np.asarray(keys)
    if bound <= (1 << 16):
        return np.argsort(keys.astype(np.uint16), kind='stable')
    order = np.argsort((keys & 0xFFFF).astype(np.uint16), kind='stable')
    shift = 16
    while (bound - 1) >> shift:
        digit = ((keys[order] >> shift) & 0xFFFF).astype(np.uint16)
        order = order[np.argsort(digit, kind='stable')]
        shift += 16
    return order


def ragged_arange(starts, counts):
    """concat([arange(s, s + c) for s, c in zip(starts, counts)]) without a Python loop."""
    counts = np.asarray(counts, dtype=np.int64)
    total = int(counts.sum())
    if total == 0:
        return np.zeros(0, np.int64)
    ends = np.cumsum(counts)
    base = np.repeat(np.asarray(starts, dtype=np.int64) - (ends - counts), counts)
    return base + np.arange(total, dtype=np.int64)


class FlatHistory(object):
    """Histories of a list of sequences in flat arrays: sequence i owns steps
    [seq_ptr[i], seq_ptr[i+1]); step k has timestamp step_t[k] and neighbour objects
    nbr_o[nbr_ptr[k]:nbr_ptr[k+1]] (the (r, o) arrays of the reference, of which the RGCN path only
    reads column 1, utils.py:153)."""
    __slots__ = ('seq_ptr', 'step_t', 'nbr_ptr', 'nbr_o')

    def __init__(self, seq_ptr, step_t, nbr_ptr, nbr_o):
        self.seq_ptr = np.asarray(seq_ptr, dtype=np.int64)
        self.step_t = np.asarray(step_t, dtype=np.int64)
        self.nbr_ptr = np.asarray(nbr_ptr, dtype=np.int64)
        self.nbr_o = np.asarray(nbr_o, dtype=np.int64)

    @classmethod
    def from_lists(cls, hist, hist_t):
        """From the reference layout: hist[i] = list of np.ndarray[k,2]; hist_t[i] = list of t."""
        if _listwalk is not None and isinstance(hist, list) and isinstance(hist_t, list):
            # one pass over the Python objects in C (csrc/listwalk.c): 2.4 -> ~0.3 ms per 1024 sequences; anything it does not
            # recognise (arrays that are not int64 [k, 2], tuples instead of lists) takes the numpy formulation below
            try:
                lens, cnt, nbr_o, step_t = (np.frombuffer(b, dtype=np.int64) for b in _listwalk.flatten(hist, hist_t))
                return cls(np.concatenate(([0], np.cumsum(lens))), step_t, np.concatenate(([0], np.cumsum(cnt))), nbr_o)
            except (TypeError, ValueError, OverflowError, BufferError):
                pass
        lens = np.fromiter(map(len, hist), dtype=np.int64, count=len(hist))
        seq_ptr = np.concatenate(([0], np.cumsum(lens)))
        steps = list(_chain.from_iterable(hist))
        if steps:
            # this conversion is half of the host time of a step driven through the reference's list API (train.py:136-137:
            # ~8.6 k step arrays per 1024 sequences): ONE concatenate over the [k, 2] arrays and one fromiter over the
            # timestamps instead of a reshape + slice + int() per step; anything irregular takes the per-step path
            cnt = np.fromiter(map(len, steps), dtype=np.int64, count=len(steps))
            try:
                big = np.concatenate(steps)
                if big.ndim != 2 or big.shape[1] != 2 or len(big) != int(cnt.sum()):
                    raise ValueError
                nbr_o = big[:, 1].astype(np.int64)
            except (ValueError, TypeError):
                nbr_o = np.concatenate([np.asarray(a).reshape(-1, 2)[:, 1] for a in steps]).astype(np.int64)
            try:
                step_t = np.fromiter(_chain.from_iterable(hist_t), dtype=np.int64, count=len(steps))
            except (ValueError, TypeError):
                step_t = np.fromiter((int(t) for ht in hist_t for t in ht), dtype=np.int64, count=len(steps))
        else:
            cnt = np.zeros(0, np.int64)
            nbr_o = np.zeros(0, np.int64)
            step_t = np.zeros(0, np.int64)
        return cls(seq_ptr, step_t, np.concatenate(([0], np.cumsum(cnt))), nbr_o)

    def __len__(self):
        return len(self.seq_ptr) - 1

    def take(self, idx):
        """Sub-batch (sequence indices `idx`) as a new FlatHistory."""
        idx = np.asarray(idx, dtype=np.int64)
        lens = self.seq_ptr[idx + 1] - self.seq_ptr[idx]
        steps = ragged_arange(self.seq_ptr[idx], lens)
        cnt = self.nbr_ptr[steps + 1] - self.nbr_ptr[steps]
        nb = ragged_arange(self.nbr_ptr[steps], cnt)
        return FlatHistory(np.concatenate(([0], np.cumsum(lens))), self.step_t[steps],
                           np.concatenate(([0], np.cumsum(cnt))), self.nbr_o[nb])


class SegPlan(object):
    """Sorted plan for renet_segment_add: rows `order[seg_ptr[u]:seg_ptr[u+1]]` all target `target[u]`."""
    __slots__ = ('order', 'seg_ptr', 'target', 'num_segments')

    @classmethod
    def host(cls, idx):
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        p = cls()
        L = _native()
        if L is not None and len(idx):
            n, bound = len(idx), int(idx.max()) + 1
            p.order = np.empty(n, np.int32)
            seg = np.empty(min(n, bound) + 1, np.int32)
            tgt = np.empty(min(n, bound), np.int32)
            u = L.renet_host_segplan(_p(idx), n, bound, _p(p.order), _p(seg), _p(tgt))
            p.seg_ptr, p.target, p.num_segments = seg[:u + 1].copy(), tgt[:u].copy(), int(u)
            return p
        p.order = stable_argsort(idx, int(idx.max()) + 1 if len(idx) else 1).astype(np.int32)
        srt = idx[p.order]
        if len(srt):
            first = np.concatenate(([True], srt[1:] != srt[:-1]))
            starts = np.nonzero(first)[0]
            p.target = srt[starts].astype(np.int32)
            p.seg_ptr = np.concatenate((starts, [len(srt)])).astype(np.int32)
        else:
            p.target = np.zeros(0, np.int32)
            p.seg_ptr = np.zeros(1, np.int32)
        p.num_segments = int(len(p.target))
        return p


class HostBatch(object):
    """Everything one direction of one training/inference batch needs, as numpy arrays."""
    INT_FIELDS = ('node_ent', 'row_ptr', 'col', 'etype', 'e_src', 'e_dst', 'chunk_ptr', 'chunk_type',
                  'type_chunk_ptr', 'subj_row', 'row_ent', 'row_rel', 'glob_row', 's_sorted', 'r_sorted',
                  'step_off', 'heavy_rows', 'heavy_rows_out', 'e_src2', 'e_dst2', 'chunk_ptr2', 'chunk_type2',
                  'type_chunk_ptr2', 'it_src', 'it_type', 'grp_ptr', 'rel_label', 'ent_label')
    PLANS = ('plan_node_ent', 'plan_subj_row', 'plan_s', 'plan_r')

    def set_edges(self, n, src, dst, et, num_types, heavy=None):
        """Directed edges (src -> dst, type et = type_s) -> the two device layouts:
        CSR by destination with relation-sorted rows (gather-SpMM forward / backward-wrt-h) and the
        relation-bucketed edge list cut into <= CHUNK-edge work items (backward-wrt-W)."""
        src = np.ascontiguousarray(src, dtype=np.int64)
        dst = np.ascontiguousarray(dst, dtype=np.int64)
        et = np.ascontiguousarray(et, dtype=np.int64)
        E = len(src)
        self.N, self.E, self.num_types = int(n), E, int(num_types)
        thr = HEAVY if heavy is None else int(heavy)
        L = _native()
        if L is not None:
            T = int(num_types)
            self.col, self.etype = np.empty(E, np.int32), np.empty(E, np.int32)
            self.row_ptr, self.norm = np.empty(int(n) + 1, np.int32), np.empty(int(n), np.float32)
            heavy = np.empty(int(n), np.int32)
            self.e_src, self.e_dst = np.empty(E, np.int32), np.empty(E, np.int32)
            self.type_chunk_ptr = np.empty(T + 1, np.int32)
            cap = E // CHUNK + T + 1
            ctype, cptr = np.empty(cap, np.int32), np.empty(cap + 1, np.int32)
            nh, nc = ctypes.c_int64(0), ctypes.c_int64(0)
            L.renet_host_edge_layouts(int(n), E, _p(src), _p(dst), _p(et), T, CHUNK, thr, _p(self.col),
                                      _p(self.etype), _p(self.row_ptr), _p(self.norm), _p(heavy),
                                      ctypes.byref(nh), _p(self.e_src), _p(self.e_dst), _p(self.type_chunk_ptr),
                                      _p(ctype), _p(cptr), ctypes.byref(nc))
            self.heavy_rows = heavy[:nh.value].copy()
            self.n_chunks = int(nc.value)
            self.chunk_type, self.chunk_ptr = ctype[:self.n_chunks].copy(), cptr[:self.n_chunks + 1].copy()
            return self
        by_type = stable_argsort(et, num_types)
        order = by_type[stable_argsort(dst[by_type], max(int(n), 1))]         # by destination, then type
        self.col = src[order].astype(np.int32)
        self.etype = et[order].astype(np.int32)
        deg = np.bincount(dst, minlength=n)
        self.row_ptr = np.concatenate(([0], np.cumsum(deg))).astype(np.int32)
        self.norm = (1.0 / np.maximum(deg, 1)).astype(np.float32)            # utils.py:89-93
        self.heavy_rows = np.nonzero(deg > thr)[0].astype(np.int32)
        order2 = by_type
        self.e_src = src[order2].astype(np.int32)
        self.e_dst = dst[order2].astype(np.int32)
        T = int(num_types)
        tc = np.bincount(et, minlength=T)
        tstart = np.concatenate(([0], np.cumsum(tc)))
        nch = (tc + CHUNK - 1) // CHUNK
        self.type_chunk_ptr = np.concatenate(([0], np.cumsum(nch))).astype(np.int32)
        ctype = np.repeat(np.arange(T, dtype=np.int64), nch)
        within = np.arange(int(nch.sum()), dtype=np.int64) - np.repeat(np.cumsum(nch) - nch, nch)
        self.chunk_type = ctype.astype(np.int32)
        self.chunk_ptr = np.concatenate((tstart[ctype] + within * CHUNK, [E])).astype(np.int32)
        self.n_chunks = int(len(ctype))
        return self

    def set_out_rows(self, n_out, src, dst, et):
        """Extra layouts for evaluating a layer only on rows [0, n_out): the hub rows among them and the
        relation-bucketed chunk list restricted to edges whose destination is < n_out (its dW)."""
        self.nA = int(n_out)
        self.E_out = int(self.row_ptr[n_out])          # edges into rows < n_out (== edges out of them: paired)
        self.heavy_rows_out = self.heavy_rows[self.heavy_rows < n_out]
        L = _native()
        if L is not None:
            src = np.ascontiguousarray(src, dtype=np.int64)
            dst = np.ascontiguousarray(dst, dtype=np.int64)
            et = np.ascontiguousarray(et, dtype=np.int64)
            E, T = len(src), int(self.num_types)
            e_src, e_dst = np.empty(E, np.int32), np.empty(E, np.int32)
            self.type_chunk_ptr2 = np.empty(T + 1, np.int32)
            cap = E // CHUNK + T + 1
            ctype, cptr = np.empty(cap, np.int32), np.empty(cap + 1, np.int32)
            nc = ctypes.c_int64(0)
            kept = L.renet_host_type_chunks(E, _p(src), _p(dst), _p(et), T, CHUNK, int(n_out), _p(e_src), _p(e_dst),
                                            _p(self.type_chunk_ptr2), _p(ctype), _p(cptr), ctypes.byref(nc))
            self.e_src2, self.e_dst2 = e_src[:kept].copy(), e_dst[:kept].copy()
            self.n_chunks2 = int(nc.value)
            self.chunk_type2, self.chunk_ptr2 = ctype[:self.n_chunks2].copy(), cptr[:self.n_chunks2 + 1].copy()
            return self
        m = np.asarray(dst) < n_out
        sub = HostBatch().set_edges(self.N, np.asarray(src)[m], np.asarray(dst)[m], np.asarray(et)[m],
                                    self.num_types)
        self.e_src2, self.e_dst2 = sub.e_src, sub.e_dst
        self.chunk_ptr2, self.chunk_type2 = sub.chunk_ptr, sub.chunk_type
        self.type_chunk_ptr2, self.n_chunks2 = sub.type_chunk_ptr, sub.n_chunks
        return self

    def set_gather_plan(self, n_out=None, heavy=None, budget=None):
        """Item stream + wave groups of the gather kernels (plan_gather_items) for this CSR; n_out = rows of the
        pruned last layer (default: all rows)."""
        heavy = HEAVY if heavy is None else int(heavy)
        budget = GROUP_ITEMS if budget is None else int(budget)
        n_out = self.N if n_out is None else int(n_out)
        self.heavy_thresh = heavy
        L = _native()
        if L is not None:
            cap = self.E + self.N
            it_src, it_type = np.empty(max(cap, 1), np.int32), np.empty(max(cap, 1), np.int32)
            grp = np.empty(self.N + 2, np.int32)
            ni, ngo = ctypes.c_int64(0), ctypes.c_int64(0)
            ng = L.renet_host_gather_items(self.N, _p(self.row_ptr), _p(self.col), _p(self.etype), heavy, budget,
                                           n_out, _p(it_src), _p(it_type), _p(grp), ctypes.byref(ni),
                                           ctypes.byref(ngo))
            if ng < 0:
                raise ValueError('gather plan needs budget + heavy + 1 <= 64')
            self.it_src, self.it_type = it_src[:ni.value].copy(), it_type[:ni.value].copy()
            self.grp_ptr, self.n_groups, self.n_groups_out = grp[:ng + 1].copy(), int(ng), int(ngo.value)
            return self
        self.it_src, self.it_type, self.grp_ptr, self.n_groups_out = plan_gather_items(
            self.row_ptr, self.col, self.etype, n_out, heavy, budget)
        self.n_groups = len(self.grp_ptr) - 1
        return self

    @classmethod
    def from_edges(cls, n, src, dst, type_s, num_rels, heavy=None, budget=None):
        """A bare batch graph from explicit edge lists (tests, benchmarks)."""
        return cls().set_edges(n, src, dst, type_s, 2 * num_rels, heavy=heavy).set_gather_plan(heavy=heavy,
                                                                                               budget=budget)


TABLE_ENTRIES = 1 << 24          # (slot, entity) lookup-table entries per chunk of _induced_edges
SPARSE_FACTS = 1 << 23           # facts of the member graphs' timestamps per call of the sparse filter


def _induced_edges(store, ti, num_ent, keys, new_id, sparse=False):
    """Node-induced edges of the member graphs (utils.py:115-131): slot c is the graph of store timestamp index
    ti[c]; keys = sorted slot * num_ent + entity of the batch's nodes, new_id their row numbers.  Returns the
    local (subject row, object row, relation) of every kept fact, slot-major in fact order.  Membership is a
    direct lookup in a (slot, entity) table, processed in slot chunks that bound the table to 2^24 entries (one
    chunk for a training batch's <= 240 timestamps, many for the per-sequence graphs of batched inference)."""
    Tb = len(ti)
    L_ = _native()
    if sparse and L_ is not None:
        # many small member graphs (batched inference): per-node walk through the store's subject index
        by_subj, subj_sorted = store.subject_index()
        tic = np.ascontiguousarray(ti, dtype=np.int64)
        tcnt = store.trip_ptr[tic + 1] - store.trip_ptr[tic]
        nid32 = np.ascontiguousarray(new_id, dtype=np.int32)
        table = _lookup_table(num_ent)
        # output capacity per call: a kept fact belongs to its slot's timestamp => slot chunks of <= SPARSE_FACTS facts
        bounds = [0]
        acc = 0
        for c in range(Tb):
            acc += int(tcnt[c])
            if acc > SPARSE_FACTS and c > bounds[-1]:
                bounds.append(c)
                acc = int(tcnt[c])
        bounds.append(Tb)
        outs = []
        for c0, c1 in zip(bounds[:-1], bounds[1:]):
            k0, k1 = np.searchsorted(keys, (c0 * num_ent, c1 * num_ent))
            cap = int(tcnt[c0:c1].sum())
            ls, lo, rr = np.empty(cap, np.int64), np.empty(cap, np.int64), np.empty(cap, np.int64)
            kc = np.ascontiguousarray(keys[k0:k1] - c0 * num_ent, dtype=np.int64)
            nc = np.ascontiguousarray(nid32[k0:k1])
            tc = np.ascontiguousarray(tic[c0:c1])
            m = L_.renet_host_filter_edges_sparse(_p(store.trip_ptr), _p(store.trip_s), _p(store.trip_r),
                                                  _p(store.trip_o), _p(by_subj), _p(subj_sorted), _p(tc), c1 - c0,
                                                  num_ent, _p(kc), _p(nc), k1 - k0, _p(table), _p(ls), _p(lo), _p(rr))
            outs.append((ls[:m].copy(), lo[:m].copy(), rr[:m].copy()))
        if len(outs) == 1:
            return outs[0]
        return tuple(np.concatenate([o[i] for o in outs]) for i in range(3))
    per = max(1, TABLE_ENTRIES // max(num_ent, 1))
    out_s, out_o, out_r = [], [], []
    nid32 = np.ascontiguousarray(new_id, dtype=np.int32)
    for c0 in range(0, Tb, per):
        c1 = min(Tb, c0 + per)
        k0, k1 = np.searchsorted(keys, (c0 * num_ent, c1 * num_ent))
        tic = np.ascontiguousarray(ti[c0:c1], dtype=np.int64)
        tcnt = store.trip_ptr[tic + 1] - store.trip_ptr[tic]
        kc = np.ascontiguousarray(keys[k0:k1] - c0 * num_ent, dtype=np.int64)
        table = _lookup_table(num_ent if L_ is not None else (c1 - c0) * num_ent)
        if L_ is not None:
            cap = int(tcnt.sum())
            ls, lo, rr = np.empty(cap, np.int64), np.empty(cap, np.int64), np.empty(cap, np.int64)
            nc = np.ascontiguousarray(nid32[k0:k1])
            m = L_.renet_host_filter_edges(_p(store.trip_ptr), _p(store.trip_s), _p(store.trip_r), _p(store.trip_o),
                                           _p(tic), c1 - c0, num_ent, _p(kc), _p(nc), k1 - k0, _p(table), _p(ls),
                                           _p(lo), _p(rr))
            out_s.append(ls[:m]); out_o.append(lo[:m]); out_r.append(rr[:m])
        else:
            flat = ragged_arange(store.trip_ptr[tic], tcnt)
            eslot = np.repeat(np.arange(c1 - c0, dtype=np.int64), tcnt)
            ks = eslot * num_ent + store.trip_s[flat]
            ko = eslot * num_ent + store.trip_o[flat]
            # two random gathers instead of two binary searches over the facts' endpoints
            table[kc] = nid32[k0:k1]
            ps, po = table[ks], table[ko]
            table[kc] = -1                                        # leave the scratch table clean
            keep = (ps >= 0) & (po >= 0)
            out_s.append(ps[keep].astype(np.int64)); out_o.append(po[keep].astype(np.int64))
            out_r.append(store.trip_r[flat][keep])
    if not out_s:
        z = np.zeros(0, np.int64)
        return z, z, z
    return np.concatenate(out_s), np.concatenate(out_o), np.concatenate(out_r)


def build_batch(store, num_ent, num_rels, s, r, fh, sort=True, glob_index=None, group=None, reverse_group=None,
                sparse=None):
    """Vectorised restatement of utils.py:209-283 (+115-131,149-181).

    store: GraphStore;  s, r: int arrays [B];  fh: FlatHistory of the B sequences;
    group: None = the reference's TRAINING semantics -- one node set per timestamp, the union over every sequence
      of the batch; an int array [B] = sequences with different group ids get SEPARATE member
      graphs per timestamp, i.e. the result of calling the reference once per group (what its inference does,
      model.py:329-352, one call per test quadruple) in one batch;
    Edge types are stored as type_s; the object-side pass (reverse, model.py:78) uses
    type_o = (type_s + R) mod 2R, which the kernels apply as `type_shift`;
    reverse_group: optional bool array indexed by group id: member graphs of those groups store the OBJECT-side
      edge types (type_s + R) mod 2R, so one launch with type_shift = 0 serves both directions (build_batch_both);
    sparse: edge filter for grouped batches -- True (default with groups) walks the store's subject index per node
      (many small member graphs), False uses the per-timestamp table pass of the training batches;
    glob_index: callable mapping an int64 array of timestamps to rows of the global-embedding matrix.
    Returns a HostBatch (numpy)."""
    s = np.asarray(s, dtype=np.int64).reshape(-1)
    r = np.asarray(r, dtype=np.int64).reshape(-1)
    B = len(s)
    # ids index scratch tables / counting-sort arrays of the native passes: an id outside [0, num_ent) or
    # [0, num_rels) (stat.txt disagreeing with the data) must be an error, not an out-of-bounds write
    if store.max_ent >= num_ent or store.max_rel >= num_rels:
        raise ValueError('graph_dict holds entity id %d / relation id %d but num_ent = %d, num_rels = %d'
                         % (store.max_ent, store.max_rel, num_ent, num_rels))
    if B and (s.min() < 0 or s.max() >= num_ent or r.min() < 0 or r.max() >= num_rels):
        raise ValueError('batch subject / relation id out of range')
    if len(fh.nbr_o) and (fh.nbr_o.min() < 0 or fh.nbr_o.max() >= num_ent):
        raise ValueError('history neighbour id out of range')
    lens_all = np.diff(fh.seq_ptr)
    if sort:
        perm = np.argsort(-lens_all, kind='stable')            # ONE permutation
    else:
        perm = np.arange(B)
    lens_sorted = lens_all[perm]
    nnz = int(np.count_nonzero(lens_sorted)) if sort else int(np.count_nonzero(lens_all))
    ln = lens_sorted[:nnz]
    if not sort and nnz and np.any(ln == 0):
        raise ValueError('unsorted batches must have their non-empty histories first (utils.py:251-254)')
    hb = HostBatch()
    hb.B, hb.nnz, hb.perm, hb.lens = B, nnz, perm, ln
    hb.num_types = 2 * num_rels
    s_sorted, r_sorted = s[perm], r[perm]
    hb.s_sorted, hb.r_sorted = s_sorted.astype(np.int32), r_sorted.astype(np.int32)
    S = int(ln.sum())
    hb.S = S
    L = int(ln[0]) if (nnz and sort) else (int(ln.max()) if nnz else 0)
    hb.L = L
    if sort is False and nnz and np.any(np.diff(ln) > 0):
        raise ValueError('packed layout needs non-increasing lengths')

    # sequence-major step list (the order of node_ids_graph / global_emb_list, utils.py:172-181,223-226)
    step_seq = np.repeat(np.arange(nnz, dtype=np.int64), ln)
    step_idx = ragged_arange(fh.seq_ptr[perm[:nnz]], ln)
    step_j = np.arange(S, dtype=np.int64) - np.repeat(np.cumsum(ln) - ln, ln)
    t_k = fh.step_t[step_idx]
    uniq_t, slot_k = np.unique(t_k, return_inverse=True)
    slot_group = None
    if group is not None:
        g_k = np.asarray(group, dtype=np.int64).reshape(-1)[perm][step_seq]
        pair, slot_k = np.unique(g_k * max(len(uniq_t), 1) + slot_k, return_inverse=True)
        slot_group = pair // max(len(uniq_t), 1)                  # the group of every (group, t) slot
        uniq_t = uniq_t[pair % max(len(uniq_t), 1)]               # ... and its timestamp
    Tb = len(uniq_t)
    hb.graph_t = uniq_t

    # node sets per timestamp: {subject} U {history objects}  (utils.py:149-156)
    # Row numbering: the rows that are read after the LAST RGCN layer -- the (subject, t) rows,
    # Aggregator.py:139-140 -- come first (rows [0, nA)), so that layer can be evaluated on a row prefix.
    # Every other node is an in-neighbour of a subject row, so layer 1 still needs all N rows.
    ncnt = fh.nbr_ptr[step_idx + 1] - fh.nbr_ptr[step_idx]
    L_ = _native()
    if L_ is not None and S:
        cap = S + int(ncnt.sum())
        keys, new_id = np.empty(cap, np.int64), np.empty(cap, np.int64)
        subj_pos = np.empty(S, np.int64)
        node_ent, node_slot = np.empty(cap, np.int32), np.empty(cap, np.int64)
        na = ctypes.c_int64(0)
        slot_c = np.ascontiguousarray(slot_k, dtype=np.int64)
        subj_c = np.ascontiguousarray(s_sorted[step_seq], dtype=np.int64)
        nb_c = np.ascontiguousarray(fh.nbr_ptr[step_idx], dtype=np.int64)
        ncnt_c = np.ascontiguousarray(ncnt, dtype=np.int64)
        nbr_o = np.ascontiguousarray(fh.nbr_o, dtype=np.int64)
        N = int(L_.renet_host_node_sets(S, _p(slot_c), _p(subj_c), _p(nb_c), _p(ncnt_c), _p(nbr_o), Tb, num_ent,
                                        _p(_lookup_table(num_ent)), _p(keys), _p(subj_pos), _p(new_id),
                                        _p(node_ent), _p(node_slot), ctypes.byref(na)))
        keys, new_id = keys[:N], new_id[:N]
        key_subj = None
        hb.N, hb.nA = N, int(na.value)
        hb.node_ent, hb.node_slot = node_ent[:N].copy(), node_slot[:N].copy()
    else:
        nb_flat = ragged_arange(fh.nbr_ptr[step_idx], ncnt)
        key_subj = slot_k * num_ent + s_sorted[step_seq]
        key_nbr = np.repeat(slot_k, ncnt) * num_ent + fh.nbr_o[nb_flat]
        keys = np.unique(np.concatenate((key_subj, key_nbr)))
        N = len(keys)
        hb.N = N
        subj_pos = np.searchsorted(keys, key_subj)
        is_a = np.zeros(N, dtype=bool)
        is_a[subj_pos] = True
        order_new = np.concatenate((np.nonzero(is_a)[0], np.nonzero(~is_a)[0]))
        new_id = np.empty(N, dtype=np.int64)
        new_id[order_new] = np.arange(N)
        hb.nA = int(is_a.sum())
        hb.node_ent = (keys % num_ent)[order_new].astype(np.int32)
        hb.node_slot = (keys // num_ent)[order_new]

    # node-induced edges of every member graph (utils.py:115-131)
    if Tb:
        ls, lo, rr = _induced_edges(store, store.index_of(uniq_t), num_ent, keys, new_id,
                                    sparse=(group is not None) if sparse is None else sparse)
    else:
        ls = lo = rr = np.zeros(0, np.int64)
    src = np.concatenate((ls, lo))
    dst = np.concatenate((lo, ls))
    et = np.concatenate((rr, rr + num_rels))                          # type_s (utils.py:76)
    if reverse_group is not None and slot_group is not None and len(et):
        rev = np.asarray(reverse_group, dtype=bool)[slot_group][np.asarray(hb.node_slot)[src]]
        et = np.where(rev, (et + num_rels) % (2 * num_rels), et)      # type_o (utils.py:75), model.py:78
    E = len(src)
    hb.E = E

    hb.set_edges(N, src, dst, et, 2 * num_rels)
    hb.set_out_rows(hb.nA, src, dst, et)
    hb.set_gather_plan(hb.nA)

    # packed (time-major) layout: row p = off[j] + i  <->  step j of sorted sequence i
    bs = (ln[None, :] > np.arange(L)[:, None]).sum(axis=1) if L else np.zeros(0, np.int64)
    off = np.concatenate(([0], np.cumsum(bs))).astype(np.int64)
    hb.batch_sizes = bs.astype(np.int64)
    hb.step_off = off.astype(np.int32)
    p_of_k = off[step_j] + step_seq
    inv = np.empty(S, dtype=np.int64)
    inv[p_of_k] = np.arange(S)
    hb.packed_from_seqmajor = inv                                    # packed row p -> seq-major k
    subj_row_k = new_id[subj_pos]
    hb.subj_row_seqmajor = subj_row_k
    hb.subj_row = subj_row_k[inv].astype(np.int32)
    hb.row_seq = step_seq[inv].astype(np.int32)
    hb.row_ent = s_sorted[step_seq[inv]].astype(np.int32)
    hb.row_rel = r_sorted[step_seq[inv]].astype(np.int32)
    hb.step_t_packed = t_k[inv]
    hb.glob_row = (glob_index(t_k[inv]) if glob_index is not None else np.zeros(S, np.int64)).astype(np.int32)

    hb.plan_node_ent = SegPlan.host(hb.node_ent)
    hb.plan_subj_row = SegPlan.host(hb.subj_row)
    hb.plan_s = SegPlan.host(hb.s_sorted)
    hb.plan_r = SegPlan.host(hb.r_sorted)
    return hb


def concat_histories(a, b):
    """FlatHistory of the sequences of `a` followed by those of `b`."""
    return FlatHistory(np.concatenate((a.seq_ptr, a.seq_ptr[-1] + b.seq_ptr[1:])),
                       np.concatenate((a.step_t, b.step_t)),
                       np.concatenate((a.nbr_ptr, a.nbr_ptr[-1] + b.nbr_ptr[1:])),
                       np.concatenate((a.nbr_o, b.nbr_o)))


def build_batch_both(store, num_ent, num_rels, s, r, o, fh_s, fh_o, glob_index=None):
    """The subject pass and the object pass of ONE training step (train.py:136-138: model(..., subject=True) +
    model(..., subject=False) over the same quadruples) as a single batch of 2B sequences: rows [0, B) are the
    subject-side sequences (entity s, history s_hist, relation embedding r), rows [B, 2B) the object-side ones
    (entity o, history o_hist, relation embedding R + r, model.py:70-78).  The two passes keep SEPARATE member
    graphs per timestamp (group 0 / group 1: the node sets of utils.py:149-156 are per call), the object-side graphs
    with type_o baked into their edge types, so every row of the merged batch sees exactly what its own pass would.
    Extra fields: rel_label (the relation id without the R offset, for the relation head's labels, model.py:98)
    and ent_label (o for subject-side rows, s for object-side rows), both in sorted order; is_obj per sorted row."""
    s = np.asarray(s, dtype=np.int64).reshape(-1)
    r = np.asarray(r, dtype=np.int64).reshape(-1)
    o = np.asarray(o, dtype=np.int64).reshape(-1)
    B = len(s)
    group = np.concatenate((np.zeros(B, np.int64), np.ones(B, np.int64)))
    hb = build_batch(store, num_ent, num_rels, np.concatenate((s, o)), np.concatenate((r, r)),
                     concat_histories(fh_s, fh_o), sort=True, glob_index=glob_index, group=group,
                     reverse_group=np.array([False, True]), sparse=False)
    is_obj = group[hb.perm] == 1
    hb.rel_label = hb.r_sorted.copy()
    hb.ent_label = np.concatenate((o, s))[hb.perm].astype(np.int32)
    hb.is_obj = is_obj
    shift = np.where(is_obj, num_rels, 0).astype(np.int32)
    hb.r_sorted = hb.r_sorted + shift                                 # rows of the FULL rel_embeds table [2R, H]
    hb.row_rel = hb.row_rel + shift[hb.row_seq]
    hb.plan_r = SegPlan.host(hb.r_sorted)
    return hb


def shard_sequences(hb, rank, world):
    """SURVEY 8e option (i), the EXACT data-parallel split of ONE reference batch: every rank builds the same batch
    (same node sets, same induced graphs, same norm -- quirk 4 makes them depend on every member of the batch) and
    keeps only its share of the SEQUENCES: sorted sequences rank, rank + world, rank + 2 world, ... (lengths stay
    non-increasing and balanced across the ranks).  The graph arrays are shared with `hb`; the packed sequence layout,
    labels and scatter plans are re-derived for the subset.  Summing the ranks' losses (each weighted b_rank / B, see
    RENet.loss_prepared_both(share=...)) and gradients reproduces the single-process batch up to fp32 summation order
    (in train mode: with ops.SHARED_GRAPH_SEEDS = True, so that the replicated RGCN layers draw the same dropout masks on
    every rank; the per-sequence sites differ by construction -- each rank owns different rows).
    The RGCN layers still run on the whole batch graph on every rank: this option trades scaling of the graph part
    for bit-comparable semantics; `parallel.shard_indices` (per-rank reference batches) is the scalable default."""
    import copy
    if world <= 1:
        return hb
    out = copy.copy(hb)                                   # graph fields shared (read-only from here on)
    pos = np.arange(rank, hb.B, world)                    # sorted positions kept by this rank
    live = pos[pos < hb.nnz]
    out.B, out.nnz = len(pos), len(live)
    out.perm = hb.perm[pos]
    ln = hb.lens[live]
    out.lens = ln
    for f in ('s_sorted', 'r_sorted', 'rel_label', 'ent_label', 'is_obj'):
        if hasattr(hb, f):
            setattr(out, f, np.ascontiguousarray(getattr(hb, f)[pos]))
    L = int(ln[0]) if len(ln) else 0
    out.L = L
    out.S = int(ln.sum())
    bs = (ln[None, :] > np.arange(L)[:, None]).sum(axis=1) if L else np.zeros(0, np.int64)
    off = np.concatenate(([0], np.cumsum(bs))).astype(np.int64)
    out.batch_sizes = bs.astype(np.int64)
    out.step_off = off.astype(np.int32)
    # new packed row (step j, subset sequence i') <- old packed row off_old[j] + live[i']
    j_of = np.repeat(np.arange(L, dtype=np.int64), bs)
    i_of = np.arange(out.S, dtype=np.int64) - np.repeat(off[:-1], bs)
    old = hb.step_off.astype(np.int64)[j_of] + live[i_of]
    for f in ('subj_row', 'row_ent', 'row_rel', 'glob_row', 'step_t_packed'):
        if hasattr(hb, f):
            setattr(out, f, np.ascontiguousarray(np.asarray(getattr(hb, f))[old]))
    out.row_seq = i_of.astype(np.int32)
    for f in ('packed_from_seqmajor', 'subj_row_seqmajor'):     # sequence-major helpers of the full batch: not kept
        if hasattr(out, f):
            delattr(out, f)
    out.plan_subj_row = SegPlan.host(out.subj_row)
    out.plan_s = SegPlan.host(out.s_sorted)
    out.plan_r = SegPlan.host(out.r_sorted)
    return out


def build_full_graphs(graph_dict, times):
    """Disjoint union of the FULL graphs of `times` (Aggregator.py:44-55 / 87-98, global model)."""
    hb = HostBatch()
    gs = [graph_dict[int(t)] for t in times]
    num_rels = gs[0].num_rels if gs else 0
    cnt = np.asarray([g.number_of_nodes() for g in gs], dtype=np.int64)
    off = np.concatenate(([0], np.cumsum(cnt)))
    hb.seg_ptr = off.astype(np.int32)
    hb.G = len(gs)
    hb.N = int(off[-1])
    hb.node_ent = (np.concatenate([g.ent for g in gs]) if gs else np.zeros(0, np.int64)).astype(np.int32)
    srcs, dsts, ets = [], [], []
    for g, o in zip(gs, off[:-1]):
        a, b, c = g.edges(False)
        srcs.append(a + o); dsts.append(b + o); ets.append(c)
    src = np.concatenate(srcs) if gs else np.zeros(0, np.int64)
    dst = np.concatenate(dsts) if gs else np.zeros(0, np.int64)
    et = np.concatenate(ets) if gs else np.zeros(0, np.int64)
    hb.set_edges(hb.N, src, dst, et, 2 * num_rels)
    hb.set_gather_plan()
    hb.plan_node_ent = SegPlan.host(hb.node_ent)
    return hb


SCALARS = ('N', 'E', 'S', 'B', 'nnz', 'L', 'n_chunks', 'n_chunks2', 'nA', 'E_out', 'num_types', 'G', 'n_groups',
           'n_groups_out', 'heavy_thresh')


class PackedBatch(object):
    """A HostBatch flattened for transport: ONE int32 buffer (every index array, 16-byte aligned slices),
    the fp32 `norm`, and a small picklable directory.  Built on the host (possibly in a worker process of
    pipeline.BatchPrefetcher); DeviceGraph uploads it with one H2D copy per buffer."""
    __slots__ = ('ints', 'norm', 'names', 'offs', 'sizes', 'plan_segments', 'scalars', 'host_small')

    def __init__(self, hb):
        ints, names, self.plan_segments = [], [], {}
        for f in HostBatch.INT_FIELDS + ('seg_ptr',):
            if hasattr(hb, f):
                names.append(f)
                ints.append(np.ascontiguousarray(getattr(hb, f), dtype=np.int32).reshape(-1))
        for pn in HostBatch.PLANS:
            if hasattr(hb, pn):
                p = getattr(hb, pn)
                self.plan_segments[pn] = p.num_segments
                for sub in ('order', 'seg_ptr', 'target'):
                    names.append(pn + '.' + sub)
                    ints.append(np.ascontiguousarray(getattr(p, sub), dtype=np.int32).reshape(-1))
        sizes = [len(a) for a in ints]
        padded = [(n + 3) & ~3 for n in sizes]
        buf = np.zeros(int(sum(padded)) + 4, dtype=np.int32)
        offs, o = [], 0
        for a, n, pn_ in zip(ints, sizes, padded):
            buf[o:o + n] = a
            offs.append(o)
            o += pn_
        self.ints, self.names, self.offs, self.sizes = buf, names, offs, sizes
        self.norm = np.ascontiguousarray(hb.norm, dtype=np.float32)
        self.scalars = {f: getattr(hb, f) for f in SCALARS if hasattr(hb, f)}
        # the few host-side arrays the model still needs after the upload (labels permutation, packing)
        self.host_small = {f: getattr(hb, f) for f in ('perm', 'lens', 'batch_sizes', 'step_off',
                                                       'packed_from_seqmajor', 'node_slot', 'graph_t')
                           if hasattr(hb, f)}


class _HostView(object):
    """What remains of a HostBatch on the consumer side of a PackedBatch."""

    def __init__(self, pb):
        self.__dict__.update(pb.scalars)
        self.__dict__.update(pb.host_small)


_async_h2d = threading.local()          # .on: per THREAD (builder / prefetch threads keep the blocking copy)


@contextlib.contextmanager
def async_uploads(on=True):
    """Scope in which h2d() stages small arrays through pinned buffers and copies asynchronously (the inference advance:
    RENet._joint_topk_many).  Off by default: pinned staging buffers come from torch's caching host allocator, and a
    loop that uploads hundreds of batches without ever synchronising (bench.py preparing its steps, the prefetch
    pipeline) finds no free cached block and pays a hipHostMalloc per upload (measured: host_build_ms 13 -> 43)."""
    old = getattr(_async_h2d, 'on', False)
    _async_h2d.on = bool(on) and _os.environ.get('RENET_ASYNC_H2D', '1') != '0'
    try:
        yield
    finally:
        _async_h2d.on = old


def h2d(a, device):
    """numpy array -> device tensor.  Inside an async_uploads() scope arrays up to 1 MiB go through a pinned staging
    buffer and are copied asynchronously, so that the host is not blocked on the stream (a copy from pageable memory
    waits for everything queued on the stream -- in the inference advance that serialised host batch building and
    device work chunk by chunk); larger arrays, and everything outside such a scope, use the plain blocking copy."""
    t = torch.from_numpy(np.ascontiguousarray(a))
    if not getattr(_async_h2d, 'on', False) or torch.device(device).type != 'cuda' or t.numel() == 0 or \
            t.numel() * t.element_size() > (1 << 20):
        return t.to(device)
    pinned = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    pinned.copy_(t)
    return pinned.to(device, non_blocking=True)


class DeviceGraph(object):
    """Device-resident view of a HostBatch / PackedBatch: ONE int32 upload + ONE float32 upload, sliced
    into views."""

    def __init__(self, hb, device):
        pb = hb if isinstance(hb, PackedBatch) else PackedBatch(hb)
        dev = h2d(pb.ints, device)
        self._buf = dev
        views = {nm: dev[o_:o_ + n] for nm, o_, n in zip(pb.names, pb.offs, pb.sizes)}
        for nm in pb.names:
            if '.' not in nm:
                setattr(self, nm, views[nm])
        for pn, nseg in pb.plan_segments.items():
            p = SegPlan()
            p.order, p.seg_ptr, p.target = views[pn + '.order'], views[pn + '.seg_ptr'], views[pn + '.target']
            p.num_segments = nseg
            setattr(self, pn, p)
        self.norm = h2d(pb.norm, device)
        self.ndata = {}                  # 'h' lives here, as on the reference's DGL graph
        self.heavy_thresh = pb.scalars.get('heavy_thresh', HEAVY)
        for f in ('heavy_rows', 'heavy_rows_out'):
            if getattr(self, f, None) is not None and getattr(self, f).numel() == 0:
                setattr(self, f, None)
        if not hasattr(self, 'heavy_rows_out'):
            self.heavy_rows_out = None
        for f, v in pb.scalars.items():
            setattr(self, f, v)
        if not hasattr(self, 'nA'):
            self.nA = getattr(self, 'N', None)
        self.host = hb if not isinstance(hb, PackedBatch) else _HostView(pb)
        self._table_items = None

    def table_items(self):
        """(it_src_t, it_type_t, col_t, e_src_t): the item stream / CSR columns / dW edge sources with every source
        row replaced by its ENTITY id, for the first RGCN layer addressed through the entity table
        (renet_hip.compose_table_items); composed on the device on first use."""
        if self._table_items is None:
            import renet_hip
            self._table_items = renet_hip.compose_table_items(self)
        return self._table_items
