"""Shared test helpers: rebuild the inputs of a golden fixture for the oracle / HIP path."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import fixtures, renet_oracle as O   # noqa: E402

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def renet_shapes(num_ent, num_rels, d):
    return {
        'rel_embeds': (2 * num_rels, d), 'ent_embeds': (num_ent, d),
        'encoder.weight_ih_l0': (3 * d, 4 * d), 'encoder.weight_hh_l0': (3 * d, d),
        'encoder.bias_ih_l0': (3 * d,), 'encoder.bias_hh_l0': (3 * d,),
        'encoder_r.weight_ih_l0': (3 * d, 3 * d), 'encoder_r.weight_hh_l0': (3 * d, d),
        'encoder_r.bias_ih_l0': (3 * d,), 'encoder_r.bias_hh_l0': (3 * d,),
        'aggregator.rgcn1.loop_weight': (d, d), 'aggregator.rgcn1.weight': (2 * num_rels, d * d // 100),
        'aggregator.rgcn2.loop_weight': (d, d), 'aggregator.rgcn2.weight': (2 * num_rels, d * d // 100),
        'linear.weight': (num_ent, 3 * d), 'linear.bias': (num_ent,),
        'linear_r.weight': (num_rels, 2 * d), 'linear_r.bias': (num_rels,),
    }


def global_shapes(num_ent, num_rels, d):
    return {
        'ent_embeds': (num_ent, d),
        'encoder_global.weight_ih_l0': (3 * d, d), 'encoder_global.weight_hh_l0': (3 * d, d),
        'encoder_global.bias_ih_l0': (3 * d,), 'encoder_global.bias_hh_l0': (3 * d,),
        'aggregator.rgcn1.loop_weight': (d, d), 'aggregator.rgcn1.weight': (2 * num_rels, d * d // 100),
        'aggregator.rgcn2.loop_weight': (d, d), 'aggregator.rgcn2.weight': (2 * num_rels, d * d // 100),
        'linear_s.weight': (num_ent, d), 'linear_s.bias': (num_ent,),
        'linear_o.weight': (num_ent, d), 'linear_o.bias': (num_ent,),
    }


def train_case(name, d):
    """Everything needed to re-run the golden training case `train_<name>_<d>.npz`."""
    gold = load_golden('train_%s_%d.npz' % (name, d))
    cfg, tr, va, te = fixtures.split_dataset(name)
    seq_len = int(gold['seq_len'])
    params = fixtures.make_params(int(gold['param_seed']), renet_shapes(cfg['num_ent'], cfg['num_rels'], d))
    times = np.unique(tr[:, 3])
    gl = fixtures.make_params(int(gold['global_seed']), {'g': (len(times), d)}, scale=0.3)['g']
    global_emb = {int(t): gl[k] for k, t in enumerate(times)}
    (s_hist, s_hist_t), (o_hist, o_hist_t), _ = O.build_histories(tr, cfg['num_ent'])
    idx = gold['batch_idx']
    cut = lambda h: [list(x[-seq_len:]) for x in h]
    hists = dict(s=(cut([s_hist[i] for i in idx]), cut([s_hist_t[i] for i in idx])),
                 o=(cut([o_hist[i] for i in idx]), cut([o_hist_t[i] for i in idx])))
    return dict(gold=gold, cfg=cfg, train=tr, valid=va, test=te, seq_len=seq_len, params=params,
                global_emb=global_emb, batch=tr[idx], hists=hists, d=d)
