"""RENet_global with the reference's interface (reference global_model.py:10-95): the global-graph
encoder that pretrain.py trains and model.RENet.predict consults at every new timestamp.
Same RGCN / GRU / GEMM kernels as the main model plus the segmented max/mean readout.
"""
import numpy as np
import torch
import torch.nn as nn

import ops
from Aggregator import RGCNAggregator_global
from model import GRU, _moded
from utils import soft_cross_entropy


class RENet_global(nn.Module):
    def __init__(self, in_dim, h_dim, num_rels, dropout=0, model=0, seq_len=10, num_k=10, maxpool=1):
        super().__init__()
        self.in_dim = in_dim
        self.h_dim = h_dim
        self.num_rels = num_rels
        self.model = model
        self.seq_len = seq_len
        self.num_k = num_k
        self.ent_embeds = nn.Parameter(torch.empty(in_dim, h_dim))
        nn.init.xavier_uniform_(self.ent_embeds, gain=nn.init.calculate_gain('relu'))
        self.dropout = nn.Dropout(dropout)
        self.encoder_global = GRU(h_dim, h_dim, batch_first=True)
        self.aggregator = RGCNAggregator_global(h_dim, dropout, in_dim, num_rels, 100, model, seq_len, maxpool)
        self.linear_s = nn.Linear(h_dim, in_dim)
        self.linear_o = nn.Linear(h_dim, in_dim)
        self.global_emb = None
        self.gemm_mode = None           # see model.RENet.gemm_mode

    def _head(self, subject):
        return (self.linear_s, False) if subject else (self.linear_o, True)

    @_moded
    def forward(self, t_list, true_prob_s, true_prob_o, graph_dict, subject=True):
        """global_model.py:35-55: soft cross-entropy between the predicted entity distribution at each
        timestamp and the empirical one (subject=True scores against true_prob_o, as the reference)."""
        linear, reverse = self._head(subject)
        true_prob = true_prob_o if subject else true_prob_s
        t_np = t_list.detach().cpu().numpy() if isinstance(t_list, torch.Tensor) else np.asarray(t_list)
        idx = np.argsort(-t_np, kind='stable')
        packed = self.aggregator(t_np[idx], self.ent_embeds, graph_dict, reverse=reverse)
        _, s_q = self.encoder_global(packed, total_rows=len(t_np))
        pred = ops.LinearFn.apply(s_q[0], linear.weight, linear.bias)
        return soft_cross_entropy(pred, torch.as_tensor(true_prob)[torch.from_numpy(idx)])

    @_moded
    def predict(self, t, graph_dict, subject=True):
        """global_model.py:79-92: (s_q[1,1,h], logits[1,1,N_ent], prob[N_ent]) for predicting at time t
        from the <= seq_len graphs strictly before t."""
        linear, reverse = self._head(subject)
        t = int(t)
        rnn_inp = self.aggregator.predict(t, self.ent_embeds, graph_dict, reverse=reverse)
        _, s_q = self.encoder_global(rnn_inp.view(1, -1, self.h_dim))
        sub = ops.LinearFn.apply(s_q[0], linear.weight, linear.bias).view(1, 1, -1)
        return s_q, sub, torch.softmax(sub.view(-1), dim=0)

    @_moded
    def get_global_emb(self, t_list, graph_dict):
        """global_model.py:57-73: {t: embedding used when predicting the step after t}."""
        out = dict()
        times = list(graph_dict.keys())
        time_unit = times[1] - times[0]
        prev_t = 0
        for t in t_list:
            if t == 0:
                continue
            emb, _, _ = self.predict(t, graph_dict)
            out[prev_t] = emb.detach()
            prev_t = t
        last = t_list[-1]
        out[last] = self.predict(last + int(time_unit), graph_dict)[0].detach()
        return out

    def update_global_emb(self, t, graph_dict):
        pass
