"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Config-scale parity cases: ONE eval-mode training step (both directions, model.py:64-104 +
train.py:136-138) at the sizes BASELINE.json's configs name, on the seeded synthetic streams of
re-net_amd/synth.py (the dataset blobs of ICEWS18 / WIKI / GDELT are not in the reference tree).

  tools/make_config_golden.py   runs the UNMODIFIED reference on these cases (build container only)
                                and commits losses + sampled tensors under tests/golden/config_*.npz
  tests/test_oracle_golden.py   oracle restatement vs those fixtures (CPU)
  tests/test_gpu_config.py      HIP path vs the fixtures and vs the oracle (GPU)
  bench.py                      `parity` field: HIP loss vs oracle loss on a bench batch

A case is fully determined by its entry below: stream shape + seed, batch = slice `step` of the
seeded permutation train.py would shuffle (np.random.RandomState(999), the same slices bench.py times),
parameters from fixtures.make_params, histories by the oracle's restatement of
data/*/get_history_graph.py:142-190 with history_len = seq_len.
"""
import os
import sys

import numpy as np

from . import fixtures, renet_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 're-net_amd')

CASES = {
    # BASELINE.json configs[1]: the bench workload
    'icews18_d200': dict(shape='ICEWS18', hidden=200, seq_len=10, batch=1024, step=3, param_seed=1801),
    # configs[2]: wider per-timestamp graphs, 24 relations
    'wiki_d200': dict(shape='WIKI', hidden=200, seq_len=10, batch=1024, step=1, param_seed=1802),
    # configs[3]: GDELT-shaped (240 relations, ~2 M facts; a 600-timestamp prefix keeps the CPU check short)
    'gdelt_d200': dict(shape='GDELT', hidden=200, seq_len=10, batch=1024, step=2, param_seed=1803, num_t=600),
    # configs[4]: n_hidden = 400 (4x4 relation blocks), seq_len = 15 (the arithmetic is fp32 here; the bf16
    # storage variant is compared with the same fixture at its own tolerance)
    'yago_d400_l15': dict(shape='YAGO', hidden=400, seq_len=15, batch=1024, step=2, param_seed=1804),
}


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


def _synth():
    if PKG not in sys.path:
        sys.path.insert(0, PKG)
    import synth                      # data generator only (no device code)
    return synth


def batch_indices(n_quads, step, batch):
    """Slice `step` of the seeded shuffle (train.py:127-130 with np.random.seed(999)): the slices bench.py
    times on one GPU (parallel.shard_indices with world = 1)."""
    perm = np.random.RandomState(999).permutation(n_quads)
    start = (step * batch) % max(n_quads - batch + 1, 1)
    return perm[start:start + batch]


def compare_packed(npz, key, arr, rel=2e-3):
    """`arr` against a fixture entry written by tools/make_golden.py:pack_tensor, with a tolerance RELATIVE to
    the tensor's own scale (gradients of a mean-over-1024 loss are tiny in absolute terms):
    max |diff| over the stored entries <= rel * max |ref|, and the Frobenius norms agree to rel.
    Returns (ok, max_abs_err, scale)."""
    arr = np.asarray(arr)
    if key in npz:
        ref = np.asarray(npz[key])
        got = arr.reshape(ref.shape)
        norm_ok = True
    else:
        ref = np.asarray(npz[key + '__samp'])
        got = arr.reshape(-1)[fixtures.sample_idx(arr.size)]
        nr = float(npz[key + '__norm'])
        norm_ok = abs(float(np.linalg.norm(arr.astype(np.float64))) - nr) <= rel * nr + 1e-12
    scale = float(np.abs(ref).max()) if ref.size else 0.0
    err = float(np.abs(got - ref).max()) if ref.size else 0.0
    return bool(err <= rel * scale + 1e-9 and norm_ok), err, scale


def make_params(spec, num_ent, num_rels):
    """Seeded parameter values.  Embeddings / RGCN weights xavier-like (model.py:19-25), GRU / Linear at
    torch's default scale U(-1/sqrt(fan), 1/sqrt(fan)) so that activations have trained-model magnitudes."""
    d = spec['hidden']
    shapes = renet_shapes(num_ent, num_rels, d)
    p = fixtures.make_params(spec['param_seed'], shapes)
    rng = np.random.RandomState(spec['param_seed'] + 1)
    k = 1.0 / np.sqrt(d)
    for name in sorted(shapes):
        if name.startswith('encoder'):
            p[name] = rng.uniform(-k, k, size=shapes[name]).astype(np.float32)
        elif name.startswith('linear'):
            fan = shapes['linear.weight'][1] if name.startswith('linear.') else shapes['linear_r.weight'][1]
            kk = 1.0 / np.sqrt(fan)
            p[name] = rng.uniform(-kk, kk, size=shapes[name]).astype(np.float32)
    return p


def build_case(name, with_lists=True, gold=None):
    """-> dict(spec, quads, num_ent, num_rels, idx, batch, params, global_emb{t: f32[D]},
    hists={'s': (hist, hist_t), 'o': (...)} in the reference's nested-list layout).
    gold: the case's fixture (tests/golden/config_<name>.npz) -- the batch's histories are then read from it
    (they were produced by O.build_histories over the whole stream when the fixture was generated; replaying
    that Python loop over 0.4-1.3 M facts costs 10-40 s) instead of being rebuilt."""
    spec = dict(CASES[name])
    synth = _synth()
    quads, num_ent, num_rels, unit = synth.make_stream(spec['shape'], seed=999, num_t=spec.get('num_t'))
    idx = batch_indices(len(quads), spec['step'], spec['batch'])
    d = spec['hidden']
    times = np.unique(quads[:, 3])
    gl = fixtures.make_params(spec['param_seed'] + 7, {'g': (len(times), d)}, scale=0.1)['g']
    case = dict(name=name, spec=spec, quads=quads, num_ent=num_ent, num_rels=num_rels, time_unit=unit, idx=idx,
                batch=quads[idx], params=make_params(spec, num_ent, num_rels),
                global_emb={int(t): gl[k] for k, t in enumerate(times)})
    if gold is not None:
        assert np.array_equal(np.asarray(gold['idx']), idx), 'fixture was generated for a different batch'
        case['hists'] = {tag: fixtures.unflatten_histories(gold['hist_%s_seq_ptr' % tag], gold['hist_%s_step_t' % tag],
                                                            gold['hist_%s_nbr_ptr' % tag], gold['hist_%s_nbr' % tag])
                         for tag in ('s', 'o')}
    elif with_lists:
        (sh, sht), (oh, oht), _ = O.build_histories(quads, num_ent, history_len=spec['seq_len'])
        case['hists'] = {'s': ([sh[i] for i in idx], [sht[i] for i in idx]),
                         'o': ([oh[i] for i in idx], [oht[i] for i in idx])}
    return case


def oracle_step(case, return_parts=False):
    """The oracle's eval-mode training step on a case: (loss_s, loss_o, params-with-grads[, parts])."""
    import torch
    spec = case['spec']
    params = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in case['params'].items()}
    ogd = O.build_graph_dict(case['quads'], case['num_rels'])
    ge = {t: torch.from_numpy(v) for t, v in case['global_emb'].items()}
    out = {}
    losses = []
    for tag in ('s', 'o'):
        h, ht = case['hists'][tag]
        loss, parts = O.renet_forward_loss(params, case['batch'], h, ht, ogd, ge, case['num_rels'],
                                           spec['seq_len'], subject=(tag == 's'), return_parts=True)
        losses.append(loss)
        out[tag] = parts
    (losses[0] + losses[1]).backward()
    if return_parts:
        return losses[0], losses[1], params, out
    return losses[0], losses[1], params


def unsort(parts, batch):
    """h_n / q_n / logits of one direction in ORIGINAL batch order (rows of empty histories: zero state)."""
    perm = np.asarray(parts['bg'].perm)
    res = {}
    for key, src in (('h_n', 's_h'), ('q_n', 's_q'), ('logits', 'ob_pred')):
        a = parts[src].detach().numpy()
        full = np.zeros_like(a)
        full[perm] = a
        res[key] = full
    return res


# ------------------------------------------------------------------------------------------------
# global model at pretrain scale (pretrain.py:72-86: ONE batch = every training timestamp, batch_size 1024 >= T)
# ------------------------------------------------------------------------------------------------
GLOBAL_CASES = {
    # ICEWS18-shaped: 240 full graphs in one RGCN pass (N ~ 230 k nodes, E ~ 740 k directed edges: the largest
    # gather-SpMM workload of the reference), max pooling (pretrain.py default --maxpool 1)
    'global_icews18_d200': dict(shape='ICEWS18', hidden=200, seq_len=10, maxpool=1, param_seed=2801),
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


def soft_targets(quads, num_ent, col):
    """Per-timestamp empirical distribution of column `col` (0 = subjects, 2 = objects): the INPUT the pretrain
    loss is computed against.  (A clean per-timestamp normalisation -- the reference's get_true_distribution has
    boundary quirks that tests/test_host_cpu.py pins separately; here it is only a seeded input of the right
    shape and sparsity.)  -> float64 [T, num_ent], rows in ascending time."""
    times, inv = np.unique(quads[:, 3], return_inverse=True)
    out = np.zeros((len(times), num_ent))
    np.add.at(out, (inv, quads[:, col]), 1.0)
    return out / out.sum(axis=1, keepdims=True)


def build_global_case(name):
    """-> dict(spec, quads, num_ent, num_rels, times, params, true_s, true_o)."""
    spec = dict(GLOBAL_CASES[name])
    synth = _synth()
    quads, num_ent, num_rels, unit = synth.make_stream(spec['shape'], seed=999, num_t=spec.get('num_t'))
    d = spec['hidden']
    params = fixtures.make_params(spec['param_seed'], global_shapes(num_ent, num_rels, d))
    rng = np.random.RandomState(spec['param_seed'] + 1)
    k = 1.0 / np.sqrt(d)
    for nm in sorted(params):
        if nm.startswith('encoder_global') or nm.startswith('linear_'):
            params[nm] = rng.uniform(-k, k, size=params[nm].shape).astype(np.float32)
    return dict(name=name, spec=spec, quads=quads, num_ent=num_ent, num_rels=num_rels, time_unit=unit,
                times=np.unique(quads[:, 3]), params=params,
                true_s=soft_targets(quads, num_ent, 0), true_o=soft_targets(quads, num_ent, 2))


def oracle_global_step(case, subject=True):
    """The oracle's eval-mode pretrain step (global_model.py:35-55) on a global case -> (loss, params-with-grads)."""
    import torch
    spec = case['spec']
    params = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in case['params'].items()}
    ogd = O.build_graph_dict(case['quads'], case['num_rels'])
    true = case['true_o'] if subject else case['true_s']                # global_model.py:38-43
    loss = O.global_forward_loss(params, case['times'], true, ogd, spec['seq_len'], subject=subject,
                                 maxpool=spec['maxpool'])
    loss.backward()
    return loss, params, ogd


# ------------------------------------------------------------------------------------------------
# inference state machine at config scale (test.py's loop, model.py:216-419): N_ent 23 033, R 256, num_k 1000
# ------------------------------------------------------------------------------------------------
EVAL_CASES = {
    # ICEWS18-shaped stream cut 216 / 12 / 12 timestamps (train / valid / test); the evaluated quadruples are the
    # first `per_t` of each of the first `n_t` validation timestamps => n_t - 1 timestamp advances, each scoring
    # 2 * num_k sampled entities with a [R, N_ent] joint distribution (model.py:229-297)
    'eval_icews18_d200': dict(shape='ICEWS18', hidden=200, seq_len=10, num_k=1000, n_train_t=216, n_valid_t=12,
                              n_t=3, per_t=6, model_seed=3801, global_seed=3802),
}


def build_eval_case(name):
    """-> dict(spec, num_ent, num_rels, train, valid, test, eval_idx (rows of valid), params, gparams).
    Score-head weights are drawn at a scale that gives PEAKED distributions (logit std ~2-3, as a trained model
    has): with the near-uniform softmax of default-initialised heads the 1000th and 1001st of 5.9 M joint
    probabilities differ by ~1e-9 relative and the top-k SET itself would be decided by fp32 summation order."""
    spec = dict(EVAL_CASES[name])
    synth = _synth()
    quads, num_ent, num_rels, unit = synth.make_stream(spec['shape'], seed=999)
    times = np.unique(quads[:, 3])
    a, b = spec['n_train_t'], spec['n_train_t'] + spec['n_valid_t']
    tr = quads[quads[:, 3] < times[a]]
    va = quads[(quads[:, 3] >= times[a]) & (quads[:, 3] < times[b])]
    te = quads[quads[:, 3] >= times[b]]
    d = spec['hidden']
    pm = fixtures.make_params(spec['model_seed'], renet_shapes(num_ent, num_rels, d))
    pg = fixtures.make_params(spec['global_seed'], global_shapes(num_ent, num_rels, d))
    rng = np.random.RandomState(spec['model_seed'] + 1)
    k = 1.0 / np.sqrt(d)
    for p_ in (pm, pg):
        for nm in sorted(p_):
            if nm.startswith('encoder'):
                p_[nm] = rng.uniform(-k, k, size=p_[nm].shape).astype(np.float32)
            elif nm.startswith('linear') and nm.endswith('weight'):
                p_[nm] = rng.uniform(-1.0, 1.0, size=p_[nm].shape).astype(np.float32)
            elif nm.startswith('linear'):
                p_[nm] = rng.uniform(-0.5, 0.5, size=p_[nm].shape).astype(np.float32)
    vt = np.unique(va[:, 3])
    eval_idx = np.concatenate([np.nonzero(va[:, 3] == vt[j])[0][:spec['per_t']] for j in range(spec['n_t'])])
    return dict(name=name, spec=spec, num_ent=num_ent, num_rels=num_rels, time_unit=unit, train=tr, valid=va, test=te,
                eval_idx=eval_idx, params=pm, gparams=pg)
