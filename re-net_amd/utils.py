"""Host utilities under the reference's names (reference utils.py), DGL-free.

What the drivers (train.py / test.py / pretrain.py) and model.py import from here:
get_total_number, load_quadruples, make_batch, make_batch2, get_true_distribution, get_big_graph,
get_data, soft_cross_entropy, cuda, move_dgl_to_cuda.  Graph objects are graph.TimeGraph.
"""
import os
from collections import defaultdict      # noqa: F401  (re-exported: model.py does `from utils import *`)

import numpy as np
import torch

from graph import TimeGraph, FlatHistory   # noqa: F401


def get_total_number(inPath, fileName):
    """First line of stat.txt -> (num_entities, num_relations)  (utils.py:8-12)."""
    with open(os.path.join(inPath, fileName), 'r') as fr:
        parts = fr.readline().split()
    return int(parts[0]), int(parts[1])


def load_quadruples(inPath, fileName, fileName2=None, fileName3=None):
    """Reads `s r o t [ignored]` lines (column order in the file is head, rel, tail, time) from up to
    three files -> (int array [n,4] = (s, r, o, t), sorted unique times)  (utils.py:15-52)."""
    chunks = []
    for name in (fileName, fileName2, fileName3):
        if name is None:
            continue
        path = os.path.join(inPath, name)
        if os.path.getsize(path) == 0:
            continue
        a = np.loadtxt(path, dtype=np.int64, usecols=(0, 1, 2, 3), ndmin=2)
        chunks.append(a)
    quads = np.concatenate(chunks) if chunks else np.zeros((0, 4), np.int64)
    return quads, np.unique(quads[:, 3])


def make_batch(a, b, c, n):
    for i in range(0, len(a), n):
        yield a[i:i + n], b[i:i + n], c[i:i + n]


def make_batch2(a, b, c, d, e, n):
    for i in range(0, len(a), n):
        yield a[i:i + n], b[i:i + n], c[i:i + n], d[i:i + n], e[i:i + n]


def get_big_graph(data, num_rels):
    """Per-timestamp multigraph from (s, r, o) rows (utils.py:68-87) as a DGL-free TimeGraph."""
    return TimeGraph.from_triples(np.asarray(data)[:, :3], num_rels)


def build_graph_dict(quads, num_rels):
    """data/*/get_history_graph.py:137-140: {t: graph} in ascending time order (insertion order is the
    timeline the global model walks, Aggregator.py:28-29)."""
    quads = np.asarray(quads, dtype=np.int64)
    order = np.argsort(quads[:, 3], kind='stable')
    q = quads[order]
    times, starts = np.unique(q[:, 3], return_index=True)
    ends = np.concatenate((starts[1:], [len(q)]))
    return {int(t): TimeGraph.from_triples(q[a:b, :3], num_rels) for t, a, b in zip(times, starts, ends)}


def get_data(s_hist, o_hist):
    """Facts held in the per-entity prediction caches -> unique (s, r, o) rows (utils.py:95-113).
    s_hist[e] is [] or an int array/tensor [k,2] of (r, o) for subject e; o_hist[e] likewise (r, s)."""
    rows = []
    for e, h in enumerate(s_hist):
        if len(h):
            a = np.asarray(h.cpu() if isinstance(h, torch.Tensor) else h, dtype=np.int64).reshape(-1, 2)
            rows.append(np.stack((np.full(len(a), e), a[:, 0], a[:, 1]), axis=1))
    for e, h in enumerate(o_hist):
        if len(h):
            a = np.asarray(h.cpu() if isinstance(h, torch.Tensor) else h, dtype=np.int64).reshape(-1, 2)
            rows.append(np.stack((a[:, 1], a[:, 0], np.full(len(a), e)), axis=1))
    if not rows:
        return None
    return np.unique(np.concatenate(rows), axis=0)


def get_true_distribution(train_data, num_s):
    """Per-timestamp empirical subject / object distributions with the reference's exact bucketing
    (utils.py:292-324): a fact is counted BEFORE the timestamp change is tested, so the first fact of
    a new timestamp lands in the previous timestamp's row, and the last row is left un-normalised."""
    train_data = np.asarray(train_data)
    rows_s, rows_o = [], []
    cur_s, cur_o = np.zeros(num_s), np.zeros(num_s)
    current_t = 0
    for s, o, t in zip(train_data[:, 0], train_data[:, 2], train_data[:, 3]):
        cur_s[s] += 1
        cur_o[o] += 1
        if current_t != t:
            rows_s.append(cur_s / cur_s.sum())
            rows_o.append(cur_o / cur_o.sum())
            cur_s, cur_o = np.zeros(num_s), np.zeros(num_s)
            current_t = t
    rows_s.append(cur_s)
    rows_o.append(cur_o)
    return np.stack(rows_s), np.stack(rows_o)


def soft_cross_entropy(pred, soft_targets):
    """mean_i sum_c -target[i,c] log_softmax(pred)[i,c] in float64 (utils.py:287-290)."""
    logp = torch.log_softmax(pred.double(), dim=1)
    return torch.mean(torch.sum(-soft_targets.to(logp.device).double() * logp, 1))


def cuda(tensor):
    return tensor if tensor.is_cuda else tensor.cuda()


def move_dgl_to_cuda(g):
    """No-op: batch graphs are uploaded once by graph.DeviceGraph."""
    return g
