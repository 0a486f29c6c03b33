#!/usr/bin/env python
"""One shape of the planes GEMM for counter runs:  python tools/planes_one.py [m n k a_tr b_tr reps]"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 're-net_amd'))
import renet_hip as K
m, n, k, a_tr, b_tr, reps = (int(x) for x in (sys.argv[1:7] + ['4096', '4096', '4096', '0', '0', '5'][len(sys.argv) - 1:]))
dev = torch.device('cuda:0')
a = torch.randn((k, m) if a_tr else (m, k), device=dev)
b = torch.randn((k, n) if b_tr else (n, k), device=dev)
pa, pb = K.pack_planes(a), K.pack_planes(b)
for _ in range(reps):
    K.gemm_planes(pa, bool(a_tr), pb, bool(b_tr), split_k=1)
    K.gemm(a, b, ta=bool(a_tr), tb=not b_tr, split_k=1)
torch.cuda.synchronize()
