"""TEST INFRASTRUCTURE ONLY -- the parity checker, never the product.

CPU restatement (numpy for the integer/graph work, differentiable torch-CPU for the float work)
of the RE-Net hot path named by BASELINE.json:north_star.  Only tests/, __graft_entry__.smoke()
and bench.py's `cpu_baseline` leg may import this file.  Every function cites the reference
file:line (relative to /root/reference) it restates.

Pinning: the reference ships NO golden vectors / KATs / tests, so this
restatement is pinned against OUTPUTS OF THE REFERENCE ITSELF: tools/make_golden.py runs the
unmodified reference modules (oracle/ref_loader.py + oracle/dgl_shim.py) in the build container
and commits inputs+outputs under tests/golden/; tests/test_oracle_*.py check this file against
those fixtures everywhere and against the live reference where /root/reference exists.

Third-party arithmetic on the path that is not in /root/reference:
  * DGL 0.4.x (README.md:38) -- graph container / subgraph / batch / sum-reduce; semantics restated
    in build_time_graph / induced_subgraph / batch_for_histories / rgcn_layer below.
  * torch.nn.GRU, torch.nn.Linear, CrossEntropyLoss (README.md:37 pins torch 1.6; the container has
    2.10) -- restated explicitly in gru_last_state / cross_entropy_mean; tests check them against
    torch's own CPU modules.

Conventions: `params` is a dict name -> tensor using the reference's state_dict keys
(rel_embeds, ent_embeds, encoder.weight_ih_l0, ..., aggregator.rgcn1.weight, linear.weight ...).
Histories use the reference layout: hist[i] = list (oldest first) of np.ndarray[k,2] = (r, o);
hist_t[i] = parallel list of timestamps.
"""
from collections import OrderedDict

import numpy as np
import torch

NUM_BASES = 100   # hard-coded in the reference: model.py:36, global_model.py:27


# --------------------------------------------------------------------------------------------
# graph format (utils.py:68-93)
# --------------------------------------------------------------------------------------------
class TimeGraph(object):
    """One per-timestamp multigraph, DGL-free: what utils.get_big_graph builds."""

    def __init__(self, ent, src, dst, type_s, type_o):
        self.ent = np.asarray(ent, dtype=np.int64)        # local node -> entity id  (ndata['id'])
        self.src = np.asarray(src, dtype=np.int64)
        self.dst = np.asarray(dst, dtype=np.int64)
        self.type_s = np.asarray(type_s, dtype=np.int64)  # edata['type_s']
        self.type_o = np.asarray(type_o, dtype=np.int64)  # edata['type_o']

    @property
    def num_nodes(self):
        return int(self.ent.shape[0])

    def norm(self):
        # utils.py:89-93 comp_deg_norm: 1 / max(in_degree, 1), multi-edges counted
        deg = np.bincount(self.dst, minlength=self.num_nodes).astype(np.float32)
        deg[deg == 0] = 1.0
        return (1.0 / deg).astype(np.float32)


def build_time_graph(triples, num_rels):
    """utils.py:68-87 get_big_graph.  triples: int array [m,3] = (s, r, o) of one timestamp."""
    triples = np.asarray(triples, dtype=np.int64).reshape(-1, 3)
    s, r, o = triples[:, 0], triples[:, 1], triples[:, 2]
    uniq, inv = np.unique(np.stack((s, o)), return_inverse=True)      # utils.py:70
    ls, lo = inv.reshape(2, -1)                                       # utils.py:71
    src = np.concatenate((ls, lo))                                    # utils.py:74
    dst = np.concatenate((lo, ls))
    type_s = np.concatenate((r, r + num_rels))                        # utils.py:76
    type_o = np.concatenate((r + num_rels, r))                        # utils.py:75
    return TimeGraph(uniq, src, dst, type_s, type_o)


def induced_subgraph(g, entities):
    """utils.py:115-131 make_subgraph: node-induced sub-multigraph on `entities` (global ids,
    all present in g), nodes relabelled in the order given, norm recomputed on the subgraph."""
    entities = np.asarray(list(entities), dtype=np.int64)
    pos = np.searchsorted(g.ent, entities)            # g.ids lookup (utils.py:119-120); g.ent sorted
    assert np.all(g.ent[pos] == entities), 'entity not in graph'
    relabel = np.full(g.num_nodes, -1, dtype=np.int64)
    relabel[pos] = np.arange(len(pos))
    keep = (relabel[g.src] >= 0) & (relabel[g.dst] >= 0)               # DGL node-induced subgraph
    return TimeGraph(entities, relabel[g.src[keep]], relabel[g.dst[keep]],
                     g.type_s[keep], g.type_o[keep])


class BatchGraph(object):
    """Disjoint union of per-timestamp induced subgraphs + the bookkeeping of utils.py:209-244."""
    pass


def sort_by_length(hist):
    """model.py:80-81 / utils.py:212-213: descending sort by history length.  The reference sorts
    twice with an unstable sort and assumes both give the same permutation; the
    restatement uses ONE stable descending permutation (the loss is invariant to tie order)."""
    lens = np.asarray([len(h) for h in hist], dtype=np.int64)
    perm = np.argsort(-lens, kind='stable')
    return lens[perm], perm


def batch_for_histories(hist, hist_t, s, graph_dict, sort=True):
    """utils.py:209-244 get_sorted_s_r_embed_rgcn (sort=True) / :246-283 get_s_r_embed_rgcn.

    Returns a BatchGraph with
      perm        permutation applied to the batch (identity when sort=False)
      lens        history length of each NON-EMPTY sequence, in batch order     (s_len_non_zero)
      ent, src, dst, type_s, type_o, norm   batched graph (node ids offset per member graph)
      graph_t     timestamp of each member graph, in order of first appearance  (utils.py:158-170)
      graph_off   node offset of each member graph (start_id, utils.py:165-168)
      subj_row    for every history step (sequence-major), the batched row of the subject at that
                  step's timestamp                                               (node_ids_graph)
      step_t      timestamp of every history step (sequence-major)      (order of global_emb_list)
    """
    s = np.asarray(s, dtype=np.int64).reshape(-1)
    if sort:
        lens, perm = sort_by_length(hist)
    else:
        lens = np.asarray([len(h) for h in hist], dtype=np.int64)
        perm = np.arange(len(hist))
        # utils.py:251-254: the unsorted variant still truncates at the number of non-empty
        # histories (it assumes they come first); keep that behaviour.
    nnz = int(np.count_nonzero(lens))
    lens_nz = lens[:nnz]
    seqs = [hist[perm[i]] for i in range(nnz)]
    seqs_t = [hist_t[perm[i]] for i in range(nnz)]
    s_perm = s[perm]

    # utils.py:149-156 get_neighs_by_t: t -> set(neighbour entities) U {subject}; dict order =
    # first appearance
    neighs = OrderedDict()
    for i in range(nnz):
        for nb, t in zip(seqs[i], seqs_t[i]):
            t = int(t)
            st = neighs.setdefault(t, set())
            st.update(np.asarray(nb)[:, 1].astype(np.int64).tolist())
            st.add(int(s_perm[i]))

    bg = BatchGraph()
    bg.perm, bg.lens = perm, lens_nz
    ents, srcs, dsts, tss, tos, norms = [], [], [], [], [], []
    bg.graph_t, bg.graph_off = [], []
    local = {}
    off = 0
    for t, st in neighs.items():                                   # utils.py:158-170
        sub = induced_subgraph(graph_dict[t], sorted(st))          # canonical node order: by entity
        bg.graph_t.append(t)
        bg.graph_off.append(off)
        local[t] = dict(zip(sub.ent.tolist(), range(sub.num_nodes)))
        ents.append(sub.ent)
        srcs.append(sub.src + off)
        dsts.append(sub.dst + off)
        tss.append(sub.type_s)
        tos.append(sub.type_o)
        norms.append(sub.norm())
        off += sub.num_nodes
    cat = lambda xs, dt: (np.concatenate(xs) if xs else np.zeros(0, dt)).astype(dt)
    bg.ent, bg.src, bg.dst = cat(ents, np.int64), cat(srcs, np.int64), cat(dsts, np.int64)
    bg.type_s, bg.type_o = cat(tss, np.int64), cat(tos, np.int64)
    bg.norm = cat(norms, np.float32)
    bg.num_nodes = off
    goff = dict(zip(bg.graph_t, bg.graph_off))
    subj_row, step_t = [], []
    for i in range(nnz):                                           # utils.py:172-181
        for t in seqs_t[i]:
            t = int(t)
            subj_row.append(local[t][int(s_perm[i])] + goff[t])
            step_t.append(t)
    bg.subj_row = np.asarray(subj_row, dtype=np.int64)
    bg.step_t = np.asarray(step_t, dtype=np.int64)
    return bg


# --------------------------------------------------------------------------------------------
# RGCN block-diagonal layer (RGCN.py:33-51, 79-94)
# --------------------------------------------------------------------------------------------
def rgcn_layer(h, src, dst, etype, norm, weight, loop_weight, relu, dropout_mask=None):
    """h[N,D]; weight[2R, nb*si*so] viewed [2R, nb, si, so] (RGCN.py:75-76,81-85);
    loop_weight[D,D].  out = act(norm * sum_{e: u->v} blockmul(h[u], W[type_e]) + h @ W_loop)."""
    n, d = h.shape
    si = d // NUM_BASES
    so = weight.shape[1] // (NUM_BASES * si)
    src = torch.as_tensor(src, dtype=torch.long)
    dst = torch.as_tensor(dst, dtype=torch.long)
    etype = torch.as_tensor(etype, dtype=torch.long)
    loop = h @ loop_weight                                              # RGCN.py:35
    if dropout_mask is not None:
        loop = loop * dropout_mask                                      # RGCN.py:36-37
    w = weight[etype].view(-1, NUM_BASES, si, so)                       # RGCN.py:81-85
    node = h[src].view(-1, NUM_BASES, si, 1)                            # RGCN.py:86
    msg = (node * w).sum(dim=2).reshape(-1, NUM_BASES * so)             # RGCN.py:87 (bmm 1xsi . sixso)
    agg = torch.zeros(n, NUM_BASES * so, dtype=h.dtype).index_add(0, dst, msg)   # fn.sum, RGCN.py:91
    agg = agg * torch.as_tensor(norm, dtype=h.dtype).view(-1, 1)        # RGCN.py:93-94
    out = agg + loop                                                    # RGCN.py:45-46
    return torch.relu(out) if relu else out                             # RGCN.py:47-48


def _drop_mask(shape, dropout, dtype):
    """nn.Dropout's multiplicative mask (train mode) for a tensor of `shape`, or None in eval mode (dropout == 0)."""
    if not dropout:
        return None
    return torch.nn.functional.dropout(torch.ones(shape, dtype=dtype), p=float(dropout), training=True)


def rgcn_two_layers(params, prefix, h0, bg_src, bg_dst, etype, norm, dropout=0.0):
    """Aggregator.py:119-122,136-137: rgcn1 (ReLU) then rgcn2 (identity), both with self loop.  dropout > 0: the train-mode
    masks of RGCN.py:36-37 on the two self-loop messages (torch's generator: statistical, not bit, parity with a device)."""
    m1 = _drop_mask(h0.shape, dropout, h0.dtype)
    h1 = rgcn_layer(h0, bg_src, bg_dst, etype, norm, params[prefix + 'rgcn1.weight'],
                    params[prefix + 'rgcn1.loop_weight'], relu=True, dropout_mask=m1)
    m2 = _drop_mask(h0.shape, dropout, h0.dtype)
    return rgcn_layer(h1, bg_src, bg_dst, etype, norm, params[prefix + 'rgcn2.weight'],
                      params[prefix + 'rgcn2.loop_weight'], relu=False, dropout_mask=m2)


# --------------------------------------------------------------------------------------------
# GRU (torch.nn.GRU semantics; model.py:28-29,86-88,94-96) and losses
# --------------------------------------------------------------------------------------------
def gru_last_state(x_pad, lens, w_ih, w_hh, b_ih, b_hh):
    """x_pad[B, L, I] left-aligned, lens[B] (>0, descending not required).  Returns h_n[B, H]:
    every sequence's state after ITS OWN last step (packed-sequence semantics), h_0 = 0.
    Gate order (r, z, n):  r = sig(W_ir x + b_ir + W_hr h + b_hr), z likewise,
    n = tanh(W_in x + b_in + r * (W_hn h + b_hn)),  h' = (1 - z) * n + z * h."""
    b, l, _ = x_pad.shape
    hdim = w_hh.shape[1]
    h = torch.zeros(b, hdim, dtype=x_pad.dtype)
    lens_t = torch.as_tensor(np.asarray(lens), dtype=torch.long)
    for j in range(l):
        gi = x_pad[:, j, :] @ w_ih.t() + b_ih
        gh = h @ w_hh.t() + b_hh
        i_r, i_z, i_n = gi.chunk(3, dim=1)
        h_r, h_z, h_n = gh.chunk(3, dim=1)
        r = torch.sigmoid(i_r + h_r)
        z = torch.sigmoid(i_z + h_z)
        n = torch.tanh(i_n + r * h_n)
        h_new = (1.0 - z) * n + z * h
        active = (lens_t > j).view(-1, 1).to(x_pad.dtype)
        h = active * h_new + (1.0 - active) * h
    return h


def cross_entropy_mean(logits, target):
    """nn.CrossEntropyLoss() (model.py:57): mean over rows of -log_softmax(logits)[target]."""
    target = torch.as_tensor(np.asarray(target), dtype=torch.long)
    lse = torch.logsumexp(logits, dim=1)
    return (lse - logits.gather(1, target.view(-1, 1)).view(-1)).mean()


def soft_cross_entropy(pred, soft_targets):
    """utils.py:287-290 (computed in float64)."""
    pred = pred.double()
    return torch.mean(torch.sum(-torch.as_tensor(soft_targets, dtype=torch.float64) *
                                torch.log_softmax(pred, dim=1), 1))


# --------------------------------------------------------------------------------------------
# RGCNAggregator.forward (Aggregator.py:124-167) and RENet.forward (model.py:64-104)
# --------------------------------------------------------------------------------------------
class StageTimer(object):
    """Optional wall-clock accounting per stage of the restated path (bench.py's cpu_baseline)."""

    def __init__(self):
        import time
        self._now = time.perf_counter
        self.t = {}
        self._last = self._now()

    def mark(self, stage):
        now = self._now()
        self.t[stage] = self.t.get(stage, 0.0) + (now - self._last)
        self._last = now

    def reset(self):
        self._last = self._now()


def aggregator_sequences(params, hist, hist_t, s, r, rel_embeds, graph_dict, global_emb, reverse,
                         seq_len, sort=True, timer=None, dropout=0.0):
    """Returns (bg, h2, X[B_nz, L, 4D], Xr[B_nz, L, 3D]) -- the padded tensors of Aggregator.py:144-155 before packing; in
    eval mode by default, with dropout > 0 in train mode (RGCN.py:36-37 and Aggregator.py:157-158).  global_emb: dict
    t -> tensor[..., D]."""
    if timer is not None:
        timer.reset()
    bg = batch_for_histories(hist, hist_t, s, graph_dict, sort=sort)
    if timer is not None:
        timer.mark('batch_graph')
    ent = params['ent_embeds']
    d = ent.shape[1]
    h0 = ent[torch.as_tensor(bg.ent)]                                   # utils.py:239
    etype = bg.type_o if reverse else bg.type_s                         # RGCN.py:80-85
    h2 = rgcn_two_layers(params, 'aggregator.', h0, bg.src, bg.dst, etype, bg.norm, dropout=dropout)
    if timer is not None:
        timer.mark('rgcn_x2')
    rows = h2[torch.as_tensor(bg.subj_row)]                             # Aggregator.py:139-140
    glob = torch.stack([torch.as_tensor(global_emb[int(t)]).reshape(d) for t in bg.step_t]) \
        if len(bg.step_t) else torch.zeros(0, d)
    glob = glob.to(ent.dtype)
    s_perm = np.asarray(s).reshape(-1)[bg.perm]
    r_perm = np.asarray(r).reshape(-1)[bg.perm]
    nnz = len(bg.lens)
    x = torch.zeros(nnz, seq_len, 4 * d, dtype=ent.dtype)
    xr = torch.zeros(nnz, seq_len, 3 * d, dtype=ent.dtype)
    pos = 0
    xs, xrs = [], []
    for i in range(nnz):                                                # Aggregator.py:148-155
        li = int(bg.lens[i])
        e = ent[int(s_perm[i])].view(1, d).expand(li, d)
        rr = rel_embeds[int(r_perm[i])].view(1, d).expand(li, d)
        xi = torch.cat((rows[pos:pos + li], e, rr, glob[pos:pos + li]), dim=1)
        xri = torch.cat((rows[pos:pos + li], e, glob[pos:pos + li]), dim=1)
        xs.append(torch.cat((xi, torch.zeros(seq_len - li, 4 * d, dtype=ent.dtype))))
        xrs.append(torch.cat((xri, torch.zeros(seq_len - li, 3 * d, dtype=ent.dtype))))
        pos += li
    if nnz:
        x, xr = torch.stack(xs), torch.stack(xrs)
    if dropout:                                                         # Aggregator.py:157-158 (train mode)
        x = torch.nn.functional.dropout(x, p=float(dropout), training=True)
        xr = torch.nn.functional.dropout(xr, p=float(dropout), training=True)
    if timer is not None:
        timer.mark('sequence_assembly')
    return bg, h2, x, xr


def renet_forward_loss(params, triplets, hist, hist_t, graph_dict, global_emb, num_rels, seq_len,
                       subject=True, return_parts=False, timer=None, dropout=0.0):
    """model.py:64-104; eval mode (dropout = identity) by default, train mode with dropout > 0 (the five nn.Dropout sites:
    RGCN.py:36-37 x2, Aggregator.py:157-158, model.py:90, model.py:99 -- torch's generator).  triplets: int array
    [B, >=3] (s, r, o)."""
    triplets = np.asarray(triplets, dtype=np.int64)
    if subject:                                                         # model.py:65-71
        rel_embeds = params['rel_embeds'][:num_rels]
        s, r, o = triplets[:, 0], triplets[:, 1], triplets[:, 2]
        reverse = False
    else:                                                               # model.py:72-78
        rel_embeds = params['rel_embeds'][num_rels:]
        o, r, s = triplets[:, 0], triplets[:, 1], triplets[:, 2]
        reverse = True
    ent = params['ent_embeds']
    d = ent.shape[1]
    b = len(s)
    bg, h2, x, xr = aggregator_sequences(params, hist, hist_t, s, r, rel_embeds, graph_dict,
                                         global_emb, reverse, seq_len, sort=True, timer=timer, dropout=dropout)
    nnz = len(bg.lens)
    pad = torch.zeros(b - nnz, d, dtype=ent.dtype)                      # model.py:88
    s_h = gru_last_state(x, bg.lens, params['encoder.weight_ih_l0'], params['encoder.weight_hh_l0'],
                         params['encoder.bias_ih_l0'], params['encoder.bias_hh_l0'])      # model.py:86
    s_h = torch.cat((s_h, pad), dim=0)
    if timer is not None:
        timer.mark('gru')
    sp = torch.as_tensor(s[bg.perm])
    rp = torch.as_tensor(r[bg.perm])
    feat = torch.cat((ent[sp], s_h, rel_embeds[rp]), dim=1)             # model.py:89-90
    if dropout:
        feat = torch.nn.functional.dropout(feat, p=float(dropout), training=True)       # model.py:90 (train mode)
    ob_pred = feat @ params['linear.weight'].t() + params['linear.bias']
    loss_sub = cross_entropy_mean(ob_pred, o[bg.perm])                  # model.py:91
    if timer is not None:
        timer.mark('head_ce')
    s_q = gru_last_state(xr, bg.lens, params['encoder_r.weight_ih_l0'], params['encoder_r.weight_hh_l0'],
                         params['encoder_r.bias_ih_l0'], params['encoder_r.bias_hh_l0'])  # model.py:94
    s_q = torch.cat((s_q, pad), dim=0)
    if timer is not None:
        timer.mark('gru')
    feat_r = torch.cat((ent[sp], s_q), dim=1)                           # model.py:98-99
    if dropout:
        feat_r = torch.nn.functional.dropout(feat_r, p=float(dropout), training=True)   # model.py:99 (train mode)
    ob_pred_r = feat_r @ params['linear_r.weight'].t() + params['linear_r.bias']
    loss_r = cross_entropy_mean(ob_pred_r, r[bg.perm])                  # model.py:100
    loss = loss_sub + 0.1 * loss_r                                      # model.py:103
    if timer is not None:
        timer.mark('head_ce')
    if return_parts:
        return loss, dict(bg=bg, h2=h2, x=x, xr=xr, s_h=s_h, s_q=s_q, ob_pred=ob_pred,
                          ob_pred_r=ob_pred_r, loss_sub=loss_sub, loss_r=loss_r)
    return loss


# --------------------------------------------------------------------------------------------
# global model (Aggregator.py:27-107, global_model.py:35-92)
# --------------------------------------------------------------------------------------------
def global_pool(params, times, graph_dict, reverse, maxpool=1):
    """Aggregator.py:44-61 / 87-105: batch the FULL graphs of `times`, two RGCN layers, per-graph
    max (maxpool=1) or mean over nodes -> [len(times), D]."""
    ent = params['ent_embeds']
    ents, srcs, dsts, types, norms, counts = [], [], [], [], [], []
    off = 0
    for t in times:
        g = graph_dict[int(t)]
        ents.append(g.ent)
        srcs.append(g.src + off)
        dsts.append(g.dst + off)
        types.append(g.type_o if reverse else g.type_s)
        norms.append(g.norm())
        counts.append(g.num_nodes)
        off += g.num_nodes
    h0 = ent[torch.as_tensor(np.concatenate(ents))]
    h2 = rgcn_two_layers(params, 'aggregator.', h0, np.concatenate(srcs), np.concatenate(dsts),
                         np.concatenate(types), np.concatenate(norms))
    parts = torch.split(h2, counts)
    if maxpool == 1:
        return torch.stack([p.max(dim=0)[0] for p in parts])
    return torch.stack([p.mean(dim=0) for p in parts])


def global_windows(t_list, times, seq_len):
    """Aggregator.py:28-41 / 76-85: for every target time t, the <= seq_len graph timestamps
    strictly before it (by position in the timeline `times`, uniform spacing assumed)."""
    time_unit = times[1] - times[0]
    wins = []
    for tim in t_list:
        length = int(tim // time_unit)
        wins.append(list(times[max(0, length - seq_len):length]))
    return wins


def global_forward_loss(params, t_list, true_prob, graph_dict, seq_len, subject=True, maxpool=1):
    """global_model.py:35-55 (eval mode).  t_list: int array of target timestamps (may contain 0,
    which has no history and is dropped by Aggregator.py:32-33 then zero-padded).  true_prob:
    [len(t_list), N_ent] rows aligned with t_list."""
    t_list = np.asarray(t_list, dtype=np.int64)
    perm = np.argsort(-t_list, kind='stable')                           # global_model.py:45
    sorted_t = t_list[perm]
    nnz = int(np.count_nonzero(sorted_t))
    times = list(graph_dict.keys())
    wins = global_windows(sorted_t[:nnz], times, seq_len)
    uniq = sorted(set(t for w in wins for t in w))                      # Aggregator.py:43
    pooled = global_pool(params, uniq, graph_dict, reverse=not subject, maxpool=maxpool)
    idx = {t: k for k, t in enumerate(uniq)}
    d = params['ent_embeds'].shape[1]
    x = torch.zeros(nnz, seq_len, d, dtype=pooled.dtype)
    rows = []
    for w in wins:                                                      # Aggregator.py:64-67
        xi = pooled[torch.as_tensor([idx[t] for t in w], dtype=torch.long)]
        rows.append(torch.cat((xi, torch.zeros(seq_len - len(w), d, dtype=pooled.dtype))))
    if nnz:
        x = torch.stack(rows)
    lens = [len(w) for w in wins]
    s_q = gru_last_state(x, lens, params['encoder_global.weight_ih_l0'],
                         params['encoder_global.weight_hh_l0'], params['encoder_global.bias_ih_l0'],
                         params['encoder_global.bias_hh_l0'])           # global_model.py:49
    s_q = torch.cat((s_q, torch.zeros(len(t_list) - nnz, d, dtype=s_q.dtype)))
    name = 'linear_s' if subject else 'linear_o'
    pred = s_q @ params[name + '.weight'].t() + params[name + '.bias']  # global_model.py:52
    return soft_cross_entropy(pred, np.asarray(true_prob)[perm])        # global_model.py:53


def global_predict(params, t, graph_dict, seq_len, subject=True, maxpool=1):
    """global_model.py:79-92 + Aggregator.py:75-107: embedding for predicting at time t from the
    <= seq_len graphs whose timestamp is < t.  Returns (s_q[D], logits[N_ent])."""
    times = list(graph_dict.keys())
    idx = 0
    for tt in times:                                                    # Aggregator.py:78-82
        if tt >= t:
            break
        idx += 1
    win = times[max(0, idx - seq_len):idx]
    pooled = global_pool(params, win, graph_dict, reverse=not subject, maxpool=maxpool)
    s_q = gru_last_state(pooled.unsqueeze(0), [len(win)], params['encoder_global.weight_ih_l0'],
                         params['encoder_global.weight_hh_l0'], params['encoder_global.bias_ih_l0'],
                         params['encoder_global.bias_hh_l0'])[0]
    name = 'linear_s' if subject else 'linear_o'
    logits = params[name + '.weight'] @ s_q + params[name + '.bias']
    return s_q, logits


# --------------------------------------------------------------------------------------------
# ranking metric (model.py:384-419, train.py:176-185)
# --------------------------------------------------------------------------------------------
def filtered_rank(pred, label, filter_ids):
    """model.py:391-404: sigmoid scores; every entity in `filter_ids` (all known completions of
    the query in train+valid+test) is zeroed except the gold one; rank = #greater +
    (#equal - 1)/2 + 1."""
    score = 1.0 / (1.0 + np.exp(-np.asarray(pred, dtype=np.float64)))
    score = score.astype(np.float32)        # F.sigmoid on an fp32 tensor
    ground = score[label]
    score[np.asarray(filter_ids, dtype=np.int64)] = 0
    score[label] = ground
    return float(np.sum(score > ground) + (np.sum(score == ground) - 1.0) / 2 + 1)


def mrr_hits(ranks):
    """train.py:176-181 / test.py:141-149."""
    ranks = np.asarray(ranks, dtype=np.float64)
    out = {'mrr': float(np.mean(1.0 / ranks)), 'mr': float(np.mean(ranks))}
    for k in (1, 3, 10):
        out['hits@%d' % k] = float(np.mean(ranks <= k))
    return out


# --------------------------------------------------------------------------------------------
# history construction (data/ICEWS18/get_history_graph.py:137-190)
# --------------------------------------------------------------------------------------------
def build_histories(quads, num_ent, history_len=10, state=None):
    """Streaming restatement of data/*/get_history_graph.py:142-190: for every quadruple (in file
    order) the subject's and object's rolling window of the last <= history_len PREVIOUS
    timestamps in which it was active, each snapshot an array [[r, o], ...] in file order.
    `state` carries the rolling caches from the train split into valid/test (:206-317)."""
    quads = np.asarray(quads, dtype=np.int64)
    if state is None:
        state = dict(s_his=[[] for _ in range(num_ent)], o_his=[[] for _ in range(num_ent)],
                     s_his_t=[[] for _ in range(num_ent)], o_his_t=[[] for _ in range(num_ent)],
                     s_cache={}, o_cache={}, latest_t=0)
    S, O, ST, OT = state['s_his'], state['o_his'], state['s_his_t'], state['o_his_t']
    s_hist, o_hist, s_hist_t, o_hist_t = [], [], [], []

    def flush(cache, his, his_t):                                       # :150-168
        for e, (rows, t) in cache.items():
            if len(his[e]) >= history_len:
                his[e].pop(0)
                his_t[e].pop(0)
            his[e].append(np.asarray(rows, dtype=np.int64))
            his_t[e].append(t)
        cache.clear()

    for q in quads:
        s, r, o, t = int(q[0]), int(q[1]), int(q[2]), int(q[3])
        if state['latest_t'] != t:                                      # :148
            flush(state['s_cache'], S, ST)
            flush(state['o_cache'], O, OT)
            state['latest_t'] = t
        s_hist.append(list(S[s]))                                       # :173-176
        o_hist.append(list(O[o]))
        s_hist_t.append(list(ST[s]))
        o_hist_t.append(list(OT[o]))
        rows, _ = state['s_cache'].get(s, ([], None))                   # :179-183
        state['s_cache'][s] = (rows + [[r, o]], t)
        rows, _ = state['o_cache'].get(o, ([], None))                   # :185-189
        state['o_cache'][o] = (rows + [[r, s]], t)
    return (s_hist, s_hist_t), (o_hist, o_hist_t), state


def build_graph_dict(quads, num_rels):
    """data/*/get_history_graph.py:137-140: one TimeGraph per timestamp, ascending time."""
    quads = np.asarray(quads, dtype=np.int64)
    out = OrderedDict()
    for t in np.unique(quads[:, 3]):
        out[int(t)] = build_time_graph(quads[quads[:, 3] == t][:, :3], num_rels)
    return out
