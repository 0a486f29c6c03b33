#!/usr/bin/env python
"""FULL-DATA accuracy fixture (round 5): the UNMODIFIED reference modules (oracle/ref_loader.py under oracle/dgl_shim.py)
trained on CPU over ALL of public YAGO (161 540 train / 19 523 valid quadruples, /root/reference/data/YAGO) at the
reference's own defaults -- train.py:211-236 (lr 1e-3, batch 1024, num_k 1000, grad-norm 1, seq_len 10, n_hidden 200) and
pretrain.py:113-135 for the global model (lr 1e-2, batch 1024 = every training timestamp in one batch, maxpool 1) -- for as
many epochs as the CPU affords, then train.py's filtered validation (train.py:151-185).

    python tools/make_e2e_full_golden.py <dropout> <epochs> <pre_epochs> <seed> [seed ...]

dropout 0.0 -> tests/golden/e2e_yago_full_d0.npz   (deterministic given the seed: the HIP path is compared DIRECTLY,
                                                      per-epoch loss and per-quadruple ranks)
dropout 0.5 -> tests/golden/e2e_yago_full_drop.npz (the reference's default; masks differ between torch-CPU and the
                                                      kernels' counters, so the comparison is paired by seed + statistical)
The data the GPU box needs (it has no /root/reference) is written once to tests/golden/yago_full.npz (int16/int32)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from oracle import ref_loader, renet_oracle as O   # noqa: E402
import make_e2e_drop_golden as G                   # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
YAGO = '/root/reference/data/YAGO'


def load_full():
    path = os.path.join(OUT, 'yago_full.npz')
    if os.path.exists(path):
        d = np.load(path)
        return d['train'].astype(np.int64), d['valid'].astype(np.int64), d['test'].astype(np.int64), int(d['num_ent']), \
            int(d['num_rels'])
    def rd(name):
        return np.loadtxt(os.path.join(YAGO, name), dtype=np.int64)[:, :4]
    tr, va, te = rd('train.txt'), rd('valid.txt'), rd('test.txt')
    with open(os.path.join(YAGO, 'stat.txt')) as f:
        ne, nr = [int(x) for x in f.read().split()[:2]]
    np.savez_compressed(path, train=tr.astype(np.int16), valid=va.astype(np.int16), test=te.astype(np.int16),
                        num_ent=ne, num_rels=nr)
    return tr, va, te, ne, nr


def main():
    dropout = float(sys.argv[1])
    epochs = int(sys.argv[2])
    pre_epochs = int(sys.argv[3])
    seeds = [int(x) for x in sys.argv[4:]]
    torch.set_num_threads(int(os.environ.get('RENET_GOLDEN_THREADS', '6')))
    G.CFG.update(dropout=dropout, num_k=1000, pre_epochs=pre_epochs, pre_batch=1024, pre_lr=1e-2)
    name = 'e2e_yago_full_d0.npz' if dropout == 0.0 else 'e2e_yago_full_drop.npz'
    ref = ref_loader.load()
    tr, va, te, num_ent, num_rels = load_full()
    assert tr.max() < 32768
    (sh, sht), (oh, oht), st = O.build_histories(tr, num_ent)
    (vsh, vsht), (voh, voht), st = O.build_histories(va, num_ent, state=st)
    (tsh, tsht), (toh, toht), st = O.build_histories(te, num_ent, state=st)
    hist = ((sh, sht), (oh, oht), (vsh, vsht), (voh, voht), (tsh, tsht), (toh, toht))
    res = []
    # the entity samples of the validation advance (model.py:225-227, 263-265: Categorical(prob).sample([num_k])) are recorded
    # per seed, so that the HIP run can replay the reference's random trajectory (the two devices' generators differ)
    Cat = torch.distributions.categorical.Categorical
    orig = Cat.sample
    drawn = []

    def rec(self, shape=torch.Size()):
        out = orig(self, shape)
        drawn.append(out.clone().numpy())
        return out
    Cat.sample = rec
    with ref_loader.cpu_mode():
        for seed in seeds:
            t0 = time.time()
            del drawn[:]
            res.append(G.run_seed(ref, seed, epochs, tr, va, te, num_ent, num_rels, hist))
            res[-1]['samples'] = np.stack(drawn).astype(np.int32) if drawn else np.zeros((0, G.CFG['num_k']), np.int32)
            print('seed %d total %.0f s' % (seed, time.time() - t0), flush=True)
            np.savez_compressed(
                os.path.join(OUT, name), seeds=np.asarray(seeds[:len(res)]), epochs=epochs,
                mrr=np.asarray([r['mrr'] for r in res]), hits=np.asarray([r['hits'] for r in res]),
                epoch_loss=np.asarray([r['epoch_loss'] for r in res]),
                pre_loss=np.asarray([r['pre_loss'] for r in res]),
                ranks=np.stack([r['ranks'] for r in res]).astype(np.int32),
                samples=np.stack([r['samples'] for r in res]),
                **{k: np.asarray(v) for k, v in G.CFG.items()})


if __name__ == '__main__':
    main()
