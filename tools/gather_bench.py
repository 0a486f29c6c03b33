#!/usr/bin/env python
"""Times renet_rgcn_gather on an ICEWS18-shaped batch graph for a sweep of hub-row thresholds (GPU only)."""
import os
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 're-net_amd'))
import graph as G
import preprocess as P
import renet_hip as K
import synth


def main():
    dev = torch.device('cuda:0')
    quads, ne, nr, unit = synth.make_stream('ICEWS18', seed=999)
    gd = P.build_graph_dict(quads, nr)
    hs = P.HistoryIndex(quads, 's', 10)
    idx = np.random.RandomState(999).permutation(len(quads))[:1024]
    d = 200
    w = torch.randn(2 * nr, 2 * d, device=dev) * 0.1
    for thr in (8, 16, 24, 32, 48, 64, 100000):
        G.HEAVY = thr
        hb = G.build_batch(G.store_for(gd), ne, nr, quads[idx, 0], quads[idx, 1], hs.take(idx), sort=True)
        g = G.DeviceGraph(hb, dev)
        deg = np.diff(hb.row_ptr)
        x = torch.randn(hb.N, d, device=dev)
        add = torch.randn(hb.N, d, device=dev)
        out = torch.empty_like(x)
        nbytes = hb.E * (d * 4 + 8) + hb.N * (d * 4 + 8) + w.numel() * 4 + hb.N * d * 4
        for tr in (False, True):
            def run():
                K.rgcn_gather(x, g.row_ptr, g.col, g.etype, None if tr else g.norm, w, 0, tr, add, 0.0, 0, not tr,
                              out, g.heavy_rows, thr)
            for _ in range(5):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 50
            print('thr %6d tr=%d N=%d E=%d maxdeg=%d heavy=%d heavy_edges=%d  %7.1f us  %7.1f GB/s (%.1f%% of 8 TB/s)' %
                  (thr, tr, hb.N, hb.E, deg.max(), len(hb.heavy_rows), int(deg[deg > thr].sum()), us,
                   nbytes / us / 1e3, nbytes / us / 1e3 / 80.0))


def main_global():
    """The global model's pass (Aggregator.py:44-57): ALL per-timestamp full graphs batched -- the largest RGCN
    workload of the reference and the one whose working set exceeds the 256 MiB Infinity Cache."""
    dev = torch.device('cuda:0')
    quads, ne, nr, unit = synth.make_stream('ICEWS18', seed=999)
    gd = P.build_graph_dict(quads, nr)
    d = 200
    w = torch.randn(2 * nr, 2 * d, device=dev) * 0.1
    for thr in (8, 16, 100000):
        G.HEAVY = thr
        hb = G.build_full_graphs(gd, list(gd.keys()))
        g = G.DeviceGraph(hb, dev)
        deg = np.diff(hb.row_ptr)
        x = torch.randn(hb.N, d, device=dev)
        add = torch.randn(hb.N, d, device=dev)
        out = torch.empty_like(x)
        nbytes = hb.E * (d * 4 + 8) + hb.N * (d * 4 + 8) + w.numel() * 4 + hb.N * d * 4
        for tr in (False, True):
            def run():
                K.rgcn_gather(x, g.row_ptr, g.col, g.etype, None if tr else g.norm, w, 0, tr, add, 0.0, 0, not tr,
                              out, g.heavy_rows, thr)
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 20
            print('GLOBAL thr %6d tr=%d N=%d E=%d maxdeg=%d heavy=%d  %7.1f us  %.0f MB  %7.1f GB/s (%.1f%% of 8 TB/s)' %
                  (thr, tr, hb.N, hb.E, deg.max(), len(hb.heavy_rows), us, nbytes / 1e6, nbytes / us / 1e3,
                   nbytes / us / 1e3 / 80.0))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'global':
        main_global()
    else:
        main()
