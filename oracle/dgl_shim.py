"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

A torch-CPU restatement of the handful of DGL 0.4.x calls the reference makes, so that the
UNMODIFIED reference modules under /root/reference can be imported and executed in a
container that has no `dgl` wheel and no GPU.

Third-party dependency being restated: DGL, pinned by the reference at `dgl-cu101 < 0.5`
(i.e. 0.4.x; /root/reference/README.md:38).  DGL's source is NOT in /root/reference; the
semantics below are DGL 0.4's published behaviour for exactly the call sites the reference has:

  call site (reference file:line)            DGL 0.4 behaviour restated here
  ---------------------------------------------------------------------------------------------
  utils.py:72-74  DGLGraph(), add_nodes,      mutable multigraph; duplicate edges are kept
                  add_edges(src, dst)
  utils.py:90     in_degrees(range(n))        counts every multi-edge
  utils.py:121    g.subgraph(nodes)           node-induced: keeps every parent edge whose two
                                              endpoints are both selected; nodes relabelled in the
                                              order given; ndata[NID] / edata[EID] = parent ids
  utils.py:237    g.to(device)                returns the graph (identity on CPU)
  utils.py:238, Aggregator.py:53,96 batch()   disjoint union, node ids offset by cumulative node
                                              counts in list order, features concatenated
  RGCN.py:91      update_all(msg, fn.sum,     message UDF on all edges, sum by destination (0 for
                  apply)                      zero in-degree), apply UDF on all nodes
  Aggregator.py:59,61,102,104                 max_nodes / mean_nodes: per-member-graph readout

Only what the reference touches is implemented.  Everything is plain torch on CPU.
"""
import types
import sys

import numpy as np
import torch

NID = '_ID'
EID = '_ID'


class _Frame(dict):
    """ndata / edata: a dict of tensors (the reference uses [], update, pop, iteration)."""


class _EdgeBatch(object):
    def __init__(self, src, data):
        self.src = src
        self.data = data


class _NodeBatch(object):
    def __init__(self, data):
        self.data = data


class DGLGraph(object):
    def __init__(self):
        self._n = 0
        self._src = torch.zeros(0, dtype=torch.long)
        self._dst = torch.zeros(0, dtype=torch.long)
        self.ndata = _Frame()
        self.edata = _Frame()
        self.batch_num_nodes = None

    # ---- construction (utils.py:72-77) -------------------------------------------------
    def add_nodes(self, n):
        self._n += int(n)

    def add_edges(self, u, v):
        u = torch.as_tensor(np.asarray(u), dtype=torch.long).view(-1)
        v = torch.as_tensor(np.asarray(v), dtype=torch.long).view(-1)
        self._src = torch.cat((self._src, u))
        self._dst = torch.cat((self._dst, v))

    # ---- queries ------------------------------------------------------------------------
    def number_of_nodes(self):
        return self._n

    def number_of_edges(self):
        return int(self._src.numel())

    def in_degrees(self, v=None):
        deg = torch.bincount(self._dst, minlength=self._n)
        if v is None:
            return deg
        return deg[torch.as_tensor(list(v), dtype=torch.long)]

    # ---- node-induced subgraph (utils.py:121) -------------------------------------------
    def subgraph(self, nodes):
        nodes = torch.as_tensor(list(nodes), dtype=torch.long).view(-1)
        relabel = torch.full((self._n,), -1, dtype=torch.long)
        relabel[nodes] = torch.arange(nodes.numel())
        keep = (relabel[self._src] >= 0) & (relabel[self._dst] >= 0)
        eid = torch.nonzero(keep).view(-1)
        sg = DGLGraph()
        sg._n = int(nodes.numel())
        sg._src = relabel[self._src[eid]]
        sg._dst = relabel[self._dst[eid]]
        sg.ndata[NID] = nodes
        sg.edata[EID] = eid
        return sg

    def to(self, device):
        return self

    # ---- message passing (RGCN.py:90-91) -------------------------------------------------
    def update_all(self, message_func, reduce_func, apply_node_func=None):
        msg_field, out_field = reduce_func
        edges = _EdgeBatch({k: v[self._src] for k, v in self.ndata.items()}, self.edata)
        m = message_func(edges)[msg_field]
        out = torch.zeros((self._n,) + tuple(m.shape[1:]), dtype=m.dtype)
        out = out.index_add(0, self._dst, m)
        self.ndata[out_field] = out
        if apply_node_func is not None:
            self.ndata.update(apply_node_func(_NodeBatch(self.ndata)))


def batch(graphs):
    bg = DGLGraph()
    counts = [g.number_of_nodes() for g in graphs]
    off = 0
    srcs, dsts = [], []
    for g, n in zip(graphs, counts):
        srcs.append(g._src + off)
        dsts.append(g._dst + off)
        off += n
    bg._n = off
    bg._src = torch.cat(srcs) if srcs else torch.zeros(0, dtype=torch.long)
    bg._dst = torch.cat(dsts) if dsts else torch.zeros(0, dtype=torch.long)
    if graphs:
        for k in graphs[0].ndata:
            if all(k in g.ndata for g in graphs):
                bg.ndata[k] = torch.cat([g.ndata[k] for g in graphs], dim=0)
        for k in graphs[0].edata:
            if all(k in g.edata for g in graphs):
                bg.edata[k] = torch.cat([g.edata[k] for g in graphs], dim=0)
    bg.batch_num_nodes = counts
    return bg


def _readout(g, field, op):
    parts = torch.split(g.ndata[field], g.batch_num_nodes if g.batch_num_nodes else [g._n])
    rows = []
    for p in parts:
        rows.append(p.max(dim=0)[0] if op == 'max' else p.mean(dim=0))
    return torch.stack(rows, dim=0)


def max_nodes(g, field):
    return _readout(g, field, 'max')


def mean_nodes(g, field):
    return _readout(g, field, 'mean')


def _fn_sum(msg, out):
    return (msg, out)


def install():
    """Register this shim as `dgl` / `dgl.function` in sys.modules (idempotent)."""
    mod = types.ModuleType('dgl')
    mod.DGLGraph = DGLGraph
    mod.batch = batch
    mod.max_nodes = max_nodes
    mod.mean_nodes = mean_nodes
    mod.NID = NID
    mod.EID = EID
    fn = types.ModuleType('dgl.function')
    fn.sum = _fn_sum
    mod.function = fn
    mod.__shim__ = True
    sys.modules['dgl'] = mod
    sys.modules['dgl.function'] = fn
    return mod
