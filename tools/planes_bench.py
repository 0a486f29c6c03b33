#!/usr/bin/env python
"""Times the planes GEMM (csrc/gemm_p6.h) against the in-loop-split bf16x6 kernels on the entity score head's three GEMMs
of the ICEWS18-shaped merged step (and a square reference shape), plus the producers (pack_planes, softmax_ce_planes vs
softmax_ce).  GPU only.  Usage: python tools/planes_bench.py [--iters 20]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 're-net_amd'))
import renet_hip as K  # noqa: E402

B, E, D3 = 2048, 23033, 600
SHAPES = [  # name, ta, tb, m, n, k, ones_col
    ('logits  NT', 0, 1, B, E, D3, False),
    ('dfeat   NN', 0, 0, B, D3, E, False),
    ('dW+db   TN', 1, 0, E, D3, B, True),
    ('Gi gru4 NT', 0, 1, 16000, 600, 800, False),
    ('dX gru4 NN', 0, 0, 16000, 600, 600, False),
    ('dWih4   TN', 1, 0, 600, 800, 16000, True),
    ('square  NN', 0, 0, 4096, 4096, 4096, False),
]


def timeit(fn, iters):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    for name, ta, tb, m, n, k, ones in SHAPES:
        a = torch.randn((k, m) if ta else (m, k), device=dev)
        b = torch.randn((n, k) if tb else (k, n), device=dev)
        out = torch.empty(m, n, device=dev)
        col = torch.empty(m, device=dev) if ones else None
        ap_ = K.pack_planes(a)
        bp = K.pack_planes(b, ones_col=ones)
        sk_old, sk_new = K.auto_split_k(m, n, k), K.auto_split_k_planes(m, n + int(ones), k)
        t_old = timeit(lambda: K.gemm(a, b, ta=bool(ta), tb=bool(tb), out=out, mode='bf16x6'), args.iters)
        ref = out.clone()
        t_new = timeit(lambda: K.gemm_planes(ap_, bp, ta=bool(ta), tb=bool(tb), out=out, col_out=col), args.iters)
        err = float((out - ref).abs().max() / ref.abs().max())
        t_pa = timeit(lambda: K.pack_planes(a), args.iters)
        t_pb = timeit(lambda: K.pack_planes(b, ones_col=ones), args.iters)
        fl = 2.0 * m * n * k
        print('%-11s M=%6d N=%6d K=%6d | in-loop s=%-3d %8.1f us %6.1f TF | planes s=%-3d %8.1f us %6.1f TF  (x%.2f) '
              '| pack A %6.1f us, B %6.1f us | max rel diff %.1e'
              % (name, m, n, k, sk_old, t_old, fl / t_old / 1e6, sk_new, t_new, fl / t_new / 1e6, t_old / t_new, t_pa,
                 t_pb, err), flush=True)
    # the producers of the CE gradient
    x = torch.randn(B, (E + 3) & ~3, device=dev)[:, :E]
    tgt = torch.randint(0, E, (B,), device=dev, dtype=torch.int32)
    xc = x.contiguous()
    t_f = timeit(lambda: K.softmax_ce(xc, tgt, 1.0 / B, True), args.iters)
    t_p = timeit(lambda: K.softmax_ce_planes(x, tgt, 1.0 / B), args.iters)
    print('softmax_ce fp32 in place %.1f us | planes out %.1f us' % (t_f, t_p))


if __name__ == '__main__':
    main()
