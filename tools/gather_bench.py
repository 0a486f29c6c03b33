#!/usr/bin/env python
"""Micro-benchmark of the RGCN gather-SpMM (RGCN.py:79-94) on the bench workload's batch graph (ICEWS18-shaped,
seed 999, B = 1024, D = 200) and at the global model's scale (all 240 full graphs), GPU only.

    python tools/gather_bench.py [batch|both|global|sweep|sweep_both] [--json out.json]

For every variant: forward over the full graph (fused norm + self-loop addend + ReLU), backward-wrt-h (transposed
blocks, in-place addend), the pruned forward (subject-row prefix) and the pruned backward -- the four launch classes
of a training step -- timed with HIP events on the launch stream,
  warm : 50 back-to-back launches on ONE buffer set (working set ~150 MB: it sits in the 256 MiB Infinity Cache);
  cold : launches rotating over enough buffer sets (x, addend, out each) that > 512 MiB are touched between two uses
         of the same set: every operand row comes from HBM.
Bytes are the ALGORITHMIC bytes of SURVEY 8d (renet_hip.gather_bytes); 'strict' drops the fused addend re-read.
`sweep` runs the plan parameters (hub threshold, group budget) and, in child processes, every RENET_GATHER_UNR.
"""
import json
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 're-net_amd'))
import graph as G            # noqa: E402
import preprocess as P       # noqa: E402
import renet_hip as K        # noqa: E402
import synth                 # noqa: E402

HBM = 8000.0                 # GB/s, MI355X_MICROARCH.md


SHAPE = os.environ.get('GATHER_BENCH_SHAPE', 'ICEWS18')       # GATHER_BENCH_SHAPE=YAGO GATHER_BENCH_D=400: config 5
DIM = int(os.environ.get('GATHER_BENCH_D', '200'))
SEQ = int(os.environ.get('GATHER_BENCH_SEQ', '10'))


def workload(kind):
    quads, ne, nr, unit = synth.make_stream(SHAPE, seed=999)
    gd = P.build_graph_dict(quads, nr)
    if kind == 'global':
        return lambda: G.build_full_graphs(gd, list(gd.keys())), nr
    hs = P.HistoryIndex(quads, 's', SEQ)
    idx = np.random.RandomState(999).permutation(len(quads))[3 * 1024:4 * 1024]      # bench.py's first timed batch
    store = G.store_for(gd)
    if kind == 'both':       # the merged batch of both passes (bench.py --passes merged, the default)
        ho = P.HistoryIndex(quads, 'o', SEQ)
        return lambda: G.build_batch_both(store, ne, nr, quads[idx, 0], quads[idx, 1], quads[idx, 2], hs.take(idx),
                                          ho.take(idx)), nr
    return lambda: G.build_batch(store, ne, nr, quads[idx, 0], quads[idx, 1], hs.take(idx), sort=True), nr


def time_launches(fn, sets, reps):
    for i in range(min(3, reps)):
        fn(sets[i % len(sets)])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(sets[i % len(sets)])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def bench_graph(hb, nr, dev, d=None, legacy=True, label=''):
    d = d or DIM
    g = G.DeviceGraph(hb, dev)
    n, nA = hb.N, getattr(hb, 'nA', hb.N)
    w = torch.randn(2 * nr, d * d // 100, device=dev) * 0.1
    per_set = 3 * n * d * 4
    n_sets = max(2, int((512 << 20) // per_set) + 2)
    sets = [dict(x=torch.randn(n, d, device=dev), ad=torch.randn(n, d, device=dev),
                 out=torch.empty(n, d, device=dev)) for _ in range(n_sets)]
    deg = np.diff(hb.row_ptr)
    res = {'label': label, 'N': int(n), 'E': int(hb.E), 'nA': int(nA), 'E_out': int(getattr(hb, 'E_out', hb.E)),
           'max_deg': int(deg.max()), 'hub_rows': int(len(hb.heavy_rows)), 'hub_edges': int(deg[deg > hb.heavy_thresh].sum()),
           'heavy': int(hb.heavy_thresh), 'groups': int(hb.n_groups), 'items': int(len(hb.it_src)), 'buffer_sets': n_sets,
           'unr': os.environ.get('RENET_GATHER_UNR', 'default'), 'variants': {}}

    def items_fwd(s):
        K.rgcn_gather_items(s['x'], g, w, 0, False, s['ad'], 0.0, 0, True, s['out'], use_norm=True)

    # layer 1 as the product runs it (ops.RGCNTableLayerFn): source rows AND the self-loop addend read through the node ->
    # entity map from two [N_ent, d] tables (cache resident) instead of [N, d] tensors
    n_ent = int(hb.node_ent.max()) + 1
    tab = torch.randn(n_ent, d, device=dev)
    tab_ad = torch.randn(n_ent, d, device=dev)

    def items_table_fwd(s):
        K.rgcn_gather_items_table(tab, g, w, 0, tab_ad, 0.0, 0, True, s['out'])

    def items_bwd(s):       # in place: addend == out, as in ops.RGCNLayerFn.backward
        K.rgcn_gather_items(s['x'], g, w, nr, True, s['out'], 0.0, 0, False, s['out'], use_norm=False)

    def items_fwd_pruned(s):
        K.rgcn_gather_items(s['x'], g, w, 0, False, s['ad'][:nA], 0.0, 0, False, s['out'][:nA], use_norm=True, pruned=True)

    def items_bwd_pruned(s):
        K.rgcn_gather_items(s['x'][:nA], g, w, nr, True, s['out'], 0.0, 0, False, s['out'], use_norm=False, pruned=True,
                            src_limit=nA, addend_rows=nA)

    def csr_fwd(s):
        K.rgcn_gather(s['x'], g.row_ptr, g.col, g.etype, g.norm, w, 0, False, s['ad'], 0.0, 0, True, s['out'],
                      g.heavy_rows, g.heavy_thresh)

    def csr_bwd(s):
        K.rgcn_gather(s['x'], g.row_ptr, g.col, g.etype, None, w, nr, True, s['out'], 0.0, 0, False, s['out'],
                      g.heavy_rows, g.heavy_thresh)

    full_b = K.gather_bytes(hb.E, n, d, w.numel(), True)
    full_strict = K.gather_bytes(hb.E, n, d, w.numel(), False)
    cases = [('items_fwd_full', items_fwd, full_b, full_strict), ('items_bwdh_full', items_bwd, full_b, full_strict),
             ('items_table_fwd', items_table_fwd, full_b, full_strict)]
    if nA < n:
        e_out = hb.E_out
        cases += [('items_fwd_pruned', items_fwd_pruned, K.gather_bytes(e_out, nA, d, w.numel(), True),
                   K.gather_bytes(e_out, nA, d, w.numel(), False)),
                  ('items_bwdh_pruned', items_bwd_pruned, K.gather_bytes(e_out, n, d, w.numel(), True),
                   K.gather_bytes(e_out, n, d, w.numel(), False))]
    if legacy:
        cases += [('csr_fwd_full', csr_fwd, full_b, full_strict), ('csr_bwdh_full', csr_bwd, full_b, full_strict)]
    for name, fn, nb, nb_strict in cases:
        warm = time_launches(fn, sets[:1], 50)
        cold = time_launches(fn, sets, max(2 * n_sets, 24))
        res['variants'][name] = {'warm_us': warm, 'cold_us': cold, 'bytes': nb, 'bytes_strict': nb_strict,
                                 'warm_frac': nb / warm / 1e3 / HBM, 'cold_frac': nb / cold / 1e3 / HBM,
                                 'warm_frac_strict': nb_strict / warm / 1e3 / HBM,
                                 'cold_frac_strict': nb_strict / cold / 1e3 / HBM}
        print('%-8s %-18s N=%d E=%d hub>%d:%d rows  warm %7.1f us (%.1f%% / strict %.1f%%)  cold %7.1f us (%.1f%% / strict %.1f%%)  %.0f MB'
              % (label, name, n, hb.E, hb.heavy_thresh, len(hb.heavy_rows), warm, 100 * nb / warm / 1e3 / HBM,
                 100 * nb_strict / warm / 1e3 / HBM, cold, 100 * nb / cold / 1e3 / HBM,
                 100 * nb_strict / cold / 1e3 / HBM, nb / 1e6), flush=True)
    return res


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith('-') else 'batch'
    out_json = sys.argv[sys.argv.index('--json') + 1] if '--json' in sys.argv else None
    dev = torch.device('cuda:0')
    results = []
    if mode in ('batch', 'global', 'both'):
        build, nr = workload(mode)
        results.append(bench_graph(build(), nr, dev, label=mode))
    elif mode in ('sweep', 'sweep_both'):
        kind = 'both' if mode == 'sweep_both' else 'batch'
        build, nr = workload(kind)
        for heavy, budget in ((4, 8), (6, 12), (8, 8), (8, 12), (8, 16), (8, 24), (12, 16), (16, 24), (24, 32)):
            G.HEAVY, G.GROUP_ITEMS = heavy, budget
            results.append(bench_graph(build(), nr, dev, legacy=(heavy == 8 and budget == 16),
                                       label='h%d_b%d' % (heavy, budget)))
        if 'RENET_GATHER_UNR' not in os.environ:
            for unr in (('2', '3', '4') if DIM == 400 else ('2', '3', '4', '6', '8')):
                r = subprocess.run([sys.executable, os.path.abspath(__file__), 'unr_child_' + kind],
                                   env=dict(os.environ, RENET_GATHER_UNR=unr), capture_output=True, text=True)
                sys.stdout.write(r.stdout)
                if r.returncode != 0:
                    sys.stdout.write(r.stderr[-2000:])
                for line in r.stdout.splitlines():
                    if line.startswith('JSON '):
                        results.append(json.loads(line[5:]))
    elif mode.startswith('unr_child'):
        build, nr = workload(mode.split('_')[-1] if mode.count('_') > 1 else 'batch')
        r = bench_graph(build(), nr, dev, legacy=False, label='unr' + os.environ.get('RENET_GATHER_UNR', '?'))
        print('JSON ' + json.dumps(r))
        return
    if out_json:
        with open(out_json, 'w') as f:
            json.dump(results, f, indent=1)


if __name__ == '__main__':
    main()
