#!/usr/bin/env python
"""Times the score head's three GEMM shapes (ICEWS18: B 1024, 3D 600, N_ent 23033) on the in-loop-split bf16x6
kernel and on the planes kernel (pack cost listed separately).   python tools/planes_bench.py"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 're-net_amd'))
import renet_hip as K


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    dev = torch.device('cuda:0')
    B, D3, N = 1024, 600, 23033
    feat = torch.randn(B, D3, device=dev)
    W = torch.randn(N, D3, device=dev) * 0.05
    dl = torch.randn(B, N, device=dev) * 0.01
    bias = torch.randn(N, device=dev)
    dW = torch.zeros(N, D3, device=dev)
    pf, pw, pd = K.pack_planes(feat), K.pack_planes(W), K.pack_planes(dl)
    gf = 2.0 * B * D3 * N / 1e6
    rows = [
        ('pack feat', lambda: K.pack_planes(feat, out=pf), None),
        ('pack W', lambda: K.pack_planes(W, out=pw), None),
        ('pack dlogits', lambda: K.pack_planes(dl, out=pd), None),
        ('logits  split', lambda: K.gemm(feat, W, tb=True, bias=bias), gf),
        ('logits  planes', lambda: K.gemm_planes(pf, False, pw, False, bias=bias), gf),
        ('dfeat   split', lambda: K.gemm(dl, W), gf),
        ('dfeat   planes', lambda: K.gemm_planes(pd, False, pw, True), gf),
        ('dW      split', lambda: K.gemm(dl, feat, ta=True, out=dW, beta=1.0), gf),
        ('dW      planes', lambda: K.gemm_planes(pd, True, pf, True, out=dW, beta=1.0), gf),
    ]
    for sk in (4, 8, 12, 16, 24):
        rows.append(('dfeat   planes sk%d' % sk, (lambda s: (lambda: K.gemm_planes(pd, False, pw, True, split_k=s)))(sk), gf))
    for name, fn, mf in rows:
        us = t(fn)
        print('%-22s %8.1f us %s' % (name, us, ('%6.1f TFLOP/s' % (mf / us)) if mf else ''), flush=True)
    # 4096^3
    a, b = torch.randn(4096, 4096, device=dev), torch.randn(4096, 4096, device=dev)
    pa, pb = K.pack_planes(a), K.pack_planes(b)
    g4 = 2.0 * 4096 ** 3 / 1e6
    for name, fn in (('4096^3 split', lambda: K.gemm(a, b, tb=True)), ('4096^3 planes', lambda: K.gemm_planes(pa, False, pb, False)),
                     ('4096^3 planes TR', lambda: K.gemm_planes(pa, True, pb, True))):
        us = t(fn, 10)
        print('%-22s %8.1f us %6.1f TFLOP/s' % (name, us, g4 / us), flush=True)


if __name__ == '__main__':
    main()
