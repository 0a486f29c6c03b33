"""DGL-free, vectorised replacement of the offline preprocessing (reference
data/<DS>/get_history_graph.py:111-317): per-timestamp graphs + rolling per-entity histories.

The reference materialises, for every quadruple, a Python list of <= history_len numpy arrays (and
re-pickles them per split).  Here the stream is indexed ONCE into a snapshot table:

  snapshot = all facts of one (entity, timestamp) pair, in file order      (the s_his_cache rows)
  snapshots of one entity are contiguous and time-ordered

and the history of quadruple i is just a RANGE of snapshot ids
  [ max(first snapshot of its entity, sid_i - history_len), sid_i )
so nothing is duplicated, `history_len` is a parameter (the seq_len=15 config needs 15, the
reference hard-codes 10 at get_history_graph.py:118), and a batch's FlatHistory is two ragged
gathers.  Semantics match the reference's streaming loop: a history holds the last <= history_len
PREVIOUS timestamps (t' < t) in which the entity was active in that role, across train -> valid ->
test in file order (the rolling state carried at :206-317).
"""
import numpy as np

from graph import FlatHistory, ragged_arange, TimeGraph


class HistoryIndex(object):
    """Histories of every quadruple of a (time-sorted) stream for one role ('s': subject histories of
    (r, o) rows; 'o': object histories of (r, s) rows)."""

    def __init__(self, quads, role, history_len=10):
        quads = np.asarray(quads, dtype=np.int64)
        if len(quads) > 1 and np.any(np.diff(quads[:, 3]) < 0):
            raise ValueError('quadruples must be sorted by time (as the dataset files are)')
        n = len(quads)
        ent = quads[:, 0] if role == 's' else quads[:, 2]
        other = quads[:, 2] if role == 's' else quads[:, 0]
        order = np.lexsort((np.arange(n), quads[:, 3], ent))          # entity, then time, then file order
        e_s, t_s = ent[order], quads[order, 3]
        new = np.ones(n, dtype=bool)
        if n:
            new[1:] = (e_s[1:] != e_s[:-1]) | (t_s[1:] != t_s[:-1])
        starts = np.nonzero(new)[0]
        self.snap_t = t_s[starts]
        self.snap_ent = e_s[starts]
        self.snap_ptr = np.concatenate((starts, [n])).astype(np.int64)
        self.nbr_r = quads[order, 1]
        self.nbr_o = other[order]
        sid_sorted = np.cumsum(new) - 1                                  # snapshot id of each sorted fact
        sid = np.empty(n, dtype=np.int64)
        sid[order] = sid_sorted
        # first snapshot of the entity owning each snapshot
        ent_new = np.ones(len(starts), dtype=bool)
        if len(starts):
            ent_new[1:] = self.snap_ent[1:] != self.snap_ent[:-1]
        first_of_ent = np.maximum.accumulate(np.where(ent_new, np.arange(len(starts)), 0))
        lo = np.maximum(first_of_ent[sid], sid - history_len)
        self.first = lo
        self.count = sid - lo
        self.history_len = history_len

    def __len__(self):
        return len(self.first)

    def take(self, idx, max_len=None):
        """FlatHistory of the quadruples `idx` (optionally keeping only the newest max_len steps)."""
        idx = np.asarray(idx, dtype=np.int64)
        cnt = self.count[idx]
        first = self.first[idx]
        if max_len is not None:
            cut = np.maximum(cnt - max_len, 0)
            first, cnt = first + cut, cnt - cut
        steps = ragged_arange(first, cnt)
        ncnt = self.snap_ptr[steps + 1] - self.snap_ptr[steps]
        nb = ragged_arange(self.snap_ptr[steps], ncnt)
        return FlatHistory(np.concatenate(([0], np.cumsum(cnt))), self.snap_t[steps],
                           np.concatenate(([0], np.cumsum(ncnt))), self.nbr_o[nb])

    def to_lists(self, idx):
        """The reference's nested layout for quadruples `idx`: (hist, hist_t) with hist[i] a list of
        int arrays [k, 2] = (r, o) -- what train_history_sub.txt / *_ob.txt unpickle to."""
        hist, hist_t = [], []
        for i in np.asarray(idx, dtype=np.int64):
            h, ht = [], []
            for k in range(self.first[i], self.first[i] + self.count[i]):
                a, b = self.snap_ptr[k], self.snap_ptr[k + 1]
                h.append(np.stack((self.nbr_r[a:b], self.nbr_o[a:b]), axis=1))
                ht.append(int(self.snap_t[k]))
            hist.append(h)
            hist_t.append(ht)
        return hist, hist_t


def build_graph_dict(quads, num_rels):
    """get_history_graph.py:137-140: {t: TimeGraph}, ascending time."""
    quads = np.asarray(quads, dtype=np.int64)
    order = np.argsort(quads[:, 3], kind='stable')
    q = quads[order]
    times, starts = np.unique(q[:, 3], return_index=True)
    ends = np.concatenate((starts[1:], [len(q)]))
    return {int(t): TimeGraph.from_triples(q[a:b, :3], num_rels) for t, a, b in zip(times, starts, ends)}


def write_reference_pickles(data_dir, history_len=10):
    """Command-line equivalent of running data/<DS>/get_history_graph.py in `data_dir`: reads stat.txt,
    train.txt, valid.txt (optional), test.txt and writes train_graphs.txt and
    {train,dev,test}_history_{sub,ob}.txt in the layout train.py:78-110 / test.py unpickle
    ([histories, timestamps] nested lists; graphs as graph.TimeGraph instead of DGLGraph)."""
    import os
    import pickle
    from utils import get_total_number, load_quadruples
    num_e, num_r = get_total_number(data_dir, 'stat.txt')
    splits = []
    for name, out in (('train.txt', 'train'), ('valid.txt', 'dev'), ('test.txt', 'test')):
        if os.path.isfile(os.path.join(data_dir, name)):
            splits.append((out, load_quadruples(data_dir, name)[0]))
    allq = np.concatenate([q for _, q in splits])
    hs, ho = HistoryIndex(allq, 's', history_len), HistoryIndex(allq, 'o', history_len)
    with open(os.path.join(data_dir, 'train_graphs.txt'), 'wb') as f:
        pickle.dump(build_graph_dict(splits[0][1], num_r), f)
    off = 0
    for out, q in splits:
        idx = np.arange(off, off + len(q))
        off += len(q)
        for tag, h in (('sub', hs), ('ob', ho)):
            with open(os.path.join(data_dir, '%s_history_%s.txt' % (out, tag)), 'wb') as f:
                pickle.dump(list(h.to_lists(idx)), f)


if __name__ == '__main__':
    import sys
    write_reference_pickles(sys.argv[1] if len(sys.argv) > 1 else '.',
                            int(sys.argv[2]) if len(sys.argv) > 2 else 10)
