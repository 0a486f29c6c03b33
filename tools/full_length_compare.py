#!/usr/bin/env python
"""Section 4 of profiles/r05_c_full_yago_parity.md: the UNMODIFIED reference trained 20 epochs on CPU (tools/make_full_length_reference.py ->
tests/golden/e2e_yago_full_len20.npz, or its .partial.npz loss curve) against the HIP runs of the same schedule (gpurun_out/r5s7/full20.json:
five seeds, seed 999 first).  Prints a markdown block.   python tools/full_length_compare.py"""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    full = os.path.join(ROOT, 'tests', 'golden', 'e2e_yago_full_len20.npz')
    part = full.replace('.npz', '.partial.npz')
    g = np.load(full if os.path.isfile(full) else part)
    ref_loss = np.asarray(g['epoch_loss'], dtype=np.float64)
    hip = json.loads(open(os.path.join(ROOT, 'gpurun_out', 'r5s7', 'full20.json')).read().strip().splitlines()[-1])['runs']
    el = np.asarray([r['epoch_loss'] for r in hip])
    n = len(ref_loss)
    print('| epoch | reference (seed 999, CPU) | HIP seed 999 | HIP mean of 5 seeds | HIP sd | reference - HIP mean, in sd |')
    print('|---:|---:|---:|---:|---:|---:|')
    for e in range(n):
        m, sd = el[:, e].mean(), el[:, e].std(ddof=1)
        print('| %d | %.4f | %.4f | %.4f | %.4f | %+.1f |' % (e + 1, ref_loss[e], el[0, e], m, sd, (ref_loss[e] - m) / sd))
    if 'test_mrr' in g.files:
        t = np.asarray([r['test_mrr'] for r in hip]); v = np.asarray([r['mrr'] for r in hip])
        th = np.asarray([r['test_hits'] for r in hip])
        print()
        print('| | valid MRR | TEST MRR | test Hits@1 / 3 / 10 |')
        print('|---|---:|---:|---|')
        print('| unmodified reference, CPU, seed 999 | %.4f | **%.4f** | %.4f / %.4f / %.4f |' % (
            float(g['valid_mrr']), float(g['test_mrr']), *[float(x) for x in g['test_hits']]))
        print('| HIP product loop, seed 999 | %.4f | %.4f | %.4f / %.4f / %.4f |' % (v[0], t[0], *th[0]))
        print('| HIP product loop, 5 seeds | %.4f +- %.4f | **%.4f +- %.4f** | %.4f / %.4f / %.4f |' % (
            v.mean(), v.std(ddof=1), t.mean(), t.std(ddof=1), *th.mean(0)))
        print('| README.md:169 | | 0.6569 | 0.6483 / 0.6632 / 0.6848 |')
        print()
        print('reference - HIP mean: test MRR %+.4f = %+.1f HIP seed standard deviations' % (
            float(g['test_mrr']) - t.mean(), (float(g['test_mrr']) - t.mean()) / t.std(ddof=1)))


if __name__ == '__main__':
    main()
