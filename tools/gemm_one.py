#!/usr/bin/env python
"""Runs one GEMM shape a few times (for rocprofv3 --pmc passes).  python tools/gemm_one.py M N K ta tb mode"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 're-net_amd'))
import renet_hip as K
m, n, k, ta, tb = [int(x) for x in sys.argv[1:6]]
mode = sys.argv[6] if len(sys.argv) > 6 else 'bf16x6'
dev = torch.device('cuda:0')
a = torch.randn((k, m) if ta else (m, k), device=dev)
b = torch.randn((n, k) if tb else (k, n), device=dev)
out = torch.empty(m, n, device=dev)
for _ in range(5):
    K.gemm(a, b, ta=bool(ta), tb=bool(tb), out=out, mode=mode, split_k=1)
torch.cuda.synchronize()
