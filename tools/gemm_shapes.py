#!/usr/bin/env python
"""Times the bf16x6 GEMM on a list of shapes: python tools/gemm_shapes.py M,N,K,ta,tb[,split] ...  (GPU only)"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 're-net_amd'))
import renet_hip as K

dev = torch.device('cuda:0')
for spec in sys.argv[1:]:
    v = [int(x) for x in spec.split(',')]
    m, n, k, ta, tb = v[:5]
    sk = v[5] if len(v) > 5 else None
    a = torch.randn((k, m) if ta else (m, k), device=dev)
    b = torch.randn((n, k) if tb else (k, n), device=dev)
    out = torch.empty(m, n, device=dev)
    for _ in range(3):
        K.gemm(a, b, ta=bool(ta), tb=bool(tb), out=out, split_k=sk)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 20
    e0.record()
    for _ in range(it):
        K.gemm(a, b, ta=bool(ta), tb=bool(tb), out=out, split_k=sk)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / it
    tiles = ((m + 127) // 128) * ((n + 127) // 128)
    print('%-28s tiles %5d ktiles %4d  %9.1f us %7.1f TF   %.3f us per k-tile round' % (
        spec, tiles, (k + 31) // 32, us, 2.0 * m * n * k / us / 1e6,
        us / ((k + 31) // 32) / max(1, -(-tiles // 256))))
