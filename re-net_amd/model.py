"""RENet with the reference's constructor, attributes, parameter names and method signatures
(reference model.py:10-446), so that train.py / test.py drive it unchanged -- computing on the HIP
kernels of librenet_hip.so instead of DGL + cuDNN/cuBLAS.

state_dict keys (checkpoint compatible): rel_embeds, ent_embeds,
encoder.{weight,bias}_{ih,hh}_l0, encoder_r.*, aggregator.rgcn{1,2}.{weight,loop_weight},
linear.{weight,bias}, linear_r.{weight,bias}.
"""
import math
from collections import defaultdict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils.rnn import PackedSequence

import graph as G
import ops
import renet_hip as K
from Aggregator import RGCNAggregator
from utils import *        # noqa: F401,F403  (the reference's model.py re-exports utils the same way)


class GRU(nn.Module):
    """Drop-in for nn.GRU(input_size, hidden_size, batch_first=True), one layer, h0 = 0, on the HIP
    recurrence.  Same parameter names/shapes/init as torch (weight_ih_l0 [3H,I], weight_hh_l0 [3H,H],
    bias_ih_l0, bias_hh_l0; U(-1/sqrt(H), 1/sqrt(H))).  Returns (None, h_n[1,B,H]): the reference only
    ever uses h_n (model.py:86-87,94-95)."""

    def __init__(self, input_size, hidden_size, batch_first=True):
        super().__init__()
        self.input_size, self.hidden_size, self.batch_first = input_size, hidden_size, batch_first
        k = 1.0 / math.sqrt(hidden_size)
        self.weight_ih_l0 = nn.Parameter(torch.empty(3 * hidden_size, input_size).uniform_(-k, k))
        self.weight_hh_l0 = nn.Parameter(torch.empty(3 * hidden_size, hidden_size).uniform_(-k, k))
        self.bias_ih_l0 = nn.Parameter(torch.empty(3 * hidden_size).uniform_(-k, k))
        self.bias_hh_l0 = nn.Parameter(torch.empty(3 * hidden_size).uniform_(-k, k))

    def forward(self, inp, total_rows=None):
        if isinstance(inp, PackedSequence):
            data, bs = inp.data, inp.batch_sizes.numpy()
        else:                                   # [B, L, I] dense, every sequence full length
            b, l, _ = inp.shape
            data = inp.transpose(0, 1).reshape(b * l, -1)
            bs = np.full(l, b, dtype=np.int64)
        off = ops.host_offsets(np.concatenate(([0], np.cumsum(bs))))
        nrows = int(bs[0]) if len(bs) else 0
        h = ops.GRUFn.apply(data, self.weight_ih_l0, self.weight_hh_l0, self.bias_ih_l0, self.bias_hh_l0,
                            off, max(total_rows or 0, nrows))
        return None, h


class RENet(nn.Module):
    def __init__(self, in_dim, h_dim, num_rels, dropout=0, model=0, seq_len=10, num_k=10):
        super().__init__()
        self.in_dim = in_dim
        self.h_dim = h_dim
        self.num_rels = num_rels
        self.model = model
        self.seq_len = seq_len
        self.num_k = num_k
        gain = nn.init.calculate_gain('relu')
        self.rel_embeds = nn.Parameter(torch.empty(2 * num_rels, h_dim))
        nn.init.xavier_uniform_(self.rel_embeds, gain=gain)
        self.ent_embeds = nn.Parameter(torch.empty(in_dim, h_dim))
        nn.init.xavier_uniform_(self.ent_embeds, gain=gain)

        self.drop_p = float(dropout or 0.0)
        self.dropout = nn.Dropout(dropout)
        self.encoder = GRU(4 * h_dim, h_dim, batch_first=True)
        self.encoder_r = GRU(3 * h_dim, h_dim, batch_first=True)
        self.aggregator = RGCNAggregator(h_dim, dropout, in_dim, num_rels, 100, model, seq_len)
        self.linear = nn.Linear(3 * h_dim, in_dim)
        self.linear_r = nn.Linear(2 * h_dim, num_rels)

        # inference-time state (saved/restored by train.py:189-195, test.py:70-81)
        self.global_emb = None
        self.s_hist_test = None
        self.o_hist_test = None
        self.s_hist_test_t = None
        self.o_hist_test_t = None
        self.s_his_cache = None
        self.o_his_cache = None
        self.s_his_cache_t = None
        self.o_his_cache_t = None
        self.graph_dict = None
        self.data = None
        self.latest_time = 0
        self._reset_candidates()
        self.criterion = nn.CrossEntropyLoss()

    def _reset_candidates(self):
        self.preds_list_s = defaultdict(lambda: torch.zeros(self.num_k))
        self.preds_ind_s = defaultdict(lambda: torch.zeros(self.num_k))
        self.preds_list_o = defaultdict(lambda: torch.zeros(self.num_k))
        self.preds_ind_o = defaultdict(lambda: torch.zeros(self.num_k))

    # ------------------------------------------------------------------------------------------
    def _direction(self, triplets, subject):
        """model.py:65-78: pick (s, r, o), the rel_embeds half and the edge-type view."""
        if subject:
            return triplets[:, 0], triplets[:, 1], triplets[:, 2], self.rel_embeds[:self.num_rels], False
        return triplets[:, 2], triplets[:, 1], triplets[:, 0], self.rel_embeds[self.num_rels:], True

    def prepare(self, triplets, hist, graph_dict, subject=True):
        """Host half of one direction of a training step: batch graph + packed layout + plans, uploaded
        once (graph.DeviceGraph).  Everything `loss_prepared` needs is device-resident afterwards, so an
        input pipeline can run this ahead of the step (bench.py does).  hist: (histories, timestamps) in
        the reference's nested-list layout, or a graph.FlatHistory."""
        trip = triplets.detach().cpu().numpy() if isinstance(triplets, torch.Tensor) else np.asarray(triplets)
        s, r, o, _, _ = self._direction(trip, subject)
        dev = self.ent_embeds.device
        prep = PreparedBatch()
        prep.subject, prep.b = bool(subject), len(s)
        g = self.aggregator.build(hist, s, r, self.ent_embeds, graph_dict, self.global_emb, sort=True)
        prep.g = g
        if g is None:       # every history empty: the reference crashes here; use h = 0
            perm = np.arange(len(s))
            prep.s_idx = torch.from_numpy(s.astype(np.int32)).to(dev)
            prep.r_idx = torch.from_numpy(r.astype(np.int32)).to(dev)
            prep.plan_s, prep.plan_r = _device_plan(s, dev), _device_plan(r, dev)
        else:
            perm = g.host.perm
            prep.s_idx, prep.r_idx, prep.plan_s, prep.plan_r = g.s_sorted, g.r_sorted, g.plan_s, g.plan_r
            prep.batch_sizes = torch.from_numpy(g.host.batch_sizes)
        prep.perm = perm
        prep.o_idx = torch.from_numpy(o[perm].astype(np.int32)).to(dev)
        return prep

    def loss_prepared(self, prep):
        """Device half (model.py:82-103): RGCN x2 -> sequence assembly -> GRU x2 -> heads -> loss."""
        subject = prep.subject
        rel_embeds = self.rel_embeds[:self.num_rels] if subject else self.rel_embeds[self.num_rels:]
        dev = self.ent_embeds.device
        b, g = prep.b, prep.g
        self.aggregator.last_batch = g
        if g is None:
            s_h = torch.zeros(b, self.h_dim, device=dev)
            s_q = torch.zeros(b, self.h_dim, device=dev)
        else:
            x, xr = self.aggregator.encode(g, self.ent_embeds, rel_embeds, reverse=not subject)
            _, s_h = self.encoder(PackedSequence(x, prep.batch_sizes), total_rows=b)      # model.py:86-88
            _, s_q = self.encoder_r(PackedSequence(xr, prep.batch_sizes), total_rows=b)   # model.py:94-96
            s_h, s_q = s_h[0], s_q[0]
        p = self.drop_p if self.training else 0.0
        loss_sub = ops.HeadCEFn.apply(self.ent_embeds, prep.s_idx, s_h, rel_embeds, prep.r_idx,
                                      self.linear.weight, self.linear.bias, prep.o_idx, prep.plan_s,
                                      prep.plan_r, p, ops.next_seed() if p > 0 else 0)   # model.py:89-91
        loss_r = ops.HeadCEFn.apply(self.ent_embeds, prep.s_idx, s_q, None, None, self.linear_r.weight,
                                    self.linear_r.bias, prep.r_idx, prep.plan_s, None, p,
                                    ops.next_seed() if p > 0 else 0)                      # model.py:98-100
        return loss_sub + 0.1 * loss_r                                                    # model.py:103

    def forward(self, triplets, s_hist, o_hist, graph_dict, subject=True):
        """Training loss of one direction (model.py:64-104): CE over objects + 0.1 * CE over relations.
        triplets: int tensor [B, >=3]; s_hist / o_hist: (histories, timestamps) in the reference's nested
        list layout, or graph.FlatHistory objects."""
        return self.loss_prepared(self.prepare(triplets, s_hist if subject else o_hist, graph_dict, subject))


class PreparedBatch(object):
    """Device-resident inputs of one direction of one step (see RENet.prepare)."""
    __slots__ = ('g', 'subject', 'b', 'perm', 's_idx', 'r_idx', 'o_idx', 'plan_s', 'plan_r', 'batch_sizes')


def _device_plan(idx, device):
    p = G.SegPlan.host(idx)
    for f in ('order', 'seg_ptr', 'target'):
        setattr(p, f, torch.from_numpy(getattr(p, f)).to(device))
    return p
