#!/usr/bin/env python
"""Times the device batch builder alone (csrc/builder.hip) on the bench workload: N builds back to back on an idle GPU.
    python tools/builder_bench.py [n_builds]      (run under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 're-net_amd'))
import gpu_builder           # noqa: E402
import parallel              # noqa: E402
import preprocess as P       # noqa: E402
import synth                 # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device('cuda:0')
    quads, ne, nr, _ = synth.make_stream('ICEWS18', seed=999)
    gd = P.build_graph_dict(quads, nr)
    hs, ho = P.HistoryIndex(quads, 's', 10), P.HistoryIndex(quads, 'o', 10)
    glob = {int(t): None for t in gd}
    ds = gpu_builder.DeviceStore(quads, hs, ho, gd, glob, ne, nr, dev)
    perm = np.random.RandomState(999).permutation(len(quads))
    for k in range(3):
        b = gpu_builder.DeviceBatch(ds, parallel.shard_indices(perm, k, 0, 1, 1024), 10)
        b.finalize()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    bs = [gpu_builder.DeviceBatch(ds, parallel.shard_indices(perm, 10 + k, 0, 1, 1024), 10) for k in range(n)]
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print('device builder: %.3f ms of GPU time per batch, %.3f ms of host (launch) time per batch; caps nodes %d edges %d'
          % (e0.elapsed_time(e1) / n, (t1 - t0) * 1e3 / n, ds.cap_nodes, ds.cap_edges))
    assert all(b.finalize() for b in bs)
    print('N %d E %d S %d' % (bs[-1].N, bs[-1].E, bs[-1].S))


if __name__ == '__main__':
    main()
