"""RGCN layers with the reference's class names and constructor signatures (reference RGCN.py:5-94),
running on the HIP kernels of librenet_hip.so.  `g` is a batch graph (graph.DeviceGraph) carrying
`g.ndata['h']`, as the DGL graph does in the reference; forward(g, reverse) replaces g.ndata['h'].

Parameter names / shapes are the reference's (checkpoint compatible):
  weight      [num_rels, num_bases * submat_in * submat_out]   (RGCN.py:75-77)
  loop_weight [in_feat, out_feat]                               (RGCN.py:19-22)
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

import ops

_RELU_FUNCS = (F.relu, torch.relu, torch.nn.functional.relu)


class RGCNLayer(nn.Module):
    def __init__(self, in_feat, out_feat, bias=None, activation=None, self_loop=False, dropout=0.0):
        super().__init__()
        if bias:
            # the reference's bias=True path cannot run (xavier_uniform_ on a 1-D tensor, RGCN.py:13-16)
            raise ValueError('bias is not supported (it is never enabled by RE-Net)')
        if not self_loop:
            raise ValueError('RE-Net always builds RGCN layers with self_loop=True (Aggregator.py:119-122)')
        if activation is not None and activation not in _RELU_FUNCS:
            raise ValueError('only ReLU / identity epilogues exist in the fused kernel')
        self.bias = None
        self.activation = activation
        self.self_loop = True
        self.drop_p = float(dropout or 0.0)
        gain = nn.init.calculate_gain('relu')
        self.loop_weight = nn.Parameter(torch.empty(in_feat, out_feat))
        nn.init.xavier_uniform_(self.loop_weight, gain=gain)


class RGCNBlockLayer(RGCNLayer):
    def __init__(self, in_feat, out_feat, num_rels, num_bases, bias=None, activation=None,
                 self_loop=False, dropout=0.0):
        super().__init__(in_feat, out_feat, bias, activation, self_loop=self_loop, dropout=dropout)
        if in_feat != out_feat or num_bases != 100 or in_feat not in (100, 200, 400):
            raise ValueError('kernels are built for num_bases=100 and n_hidden in {100, 200, 400}')
        self.num_rels = num_rels
        self.num_bases = num_bases
        self.out_feat = out_feat
        self.submat_in = in_feat // num_bases
        self.submat_out = out_feat // num_bases
        self.weight = nn.Parameter(torch.empty(num_rels, num_bases * self.submat_in * self.submat_out))
        nn.init.xavier_uniform_(self.weight, gain=nn.init.calculate_gain('relu'))

    def forward(self, g, reverse):
        p = self.drop_p if self.training else 0.0
        h_in = g.ndata['h']
        if isinstance(h_in, ops.TableRows):
            # first layer of a pass: h0 = table[idx] is never materialised (ops.RGCNTableLayerFn); beyond the 2 GiB
            # reach of the kernels' 32-bit buffer offsets fall back to the materialised rows
            if getattr(g, 'out_rows', None) is None and hasattr(g, 'grp_ptr') and \
                    max(g.N, h_in.table.shape[0]) * h_in.table.shape[1] * 4 < (1 << 31):
                g.ndata['h'] = ops.RGCNTableLayerFn.apply(h_in.table, self.weight, self.loop_weight, g, bool(reverse),
                                                          self.activation is not None, p,
                                                          ops.next_seed(graph_site=True) if p > 0 else 0)
                return g
            g.ndata['h'] = h_in.materialise()
        # g.out_rows (set by the aggregator for the LAST layer) = evaluate only the first out_rows rows
        h = ops.RGCNLayerFn.apply(g.ndata['h'], self.weight, self.loop_weight, g, bool(reverse),
                                  self.activation is not None, p, ops.next_seed(graph_site=True) if p > 0 else 0,
                                  getattr(g, 'out_rows', None))
        g.ndata['h'] = h
        return g
