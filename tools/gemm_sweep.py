#!/usr/bin/env python
"""split-K sweep of renet_gemm_f32 on the weight-gradient / dX shapes of the training step (GPU only)."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 're-net_amd'))
import renet_hip as K

N, S, B, E, D = 46075, 7624, 1024, 23033, 200
SHAPES = [('NN dfeat', 0, 0, B, 3 * D, E, (4, 6, 8, 12, 16, 24, 32, 48, 64)),
          ('TN dW_lin', 1, 0, E, 3 * D, B, (1, 2, 3, 4)),
          ('NT logits', 0, 1, B, E, 3 * D, (1, 2)),
          ('TN dW_ih4', 1, 0, 3 * D, 4 * D, S, (4, 8, 12, 14, 16, 24, 29)),
          ('TN dW_loop', 1, 0, D, D, N, (16, 32, 64, 96, 128)),
          ('TN dW_hh', 1, 0, 3 * D, D, S, (8, 16, 25, 32, 51)),
          ('NT Gi gru4', 0, 1, S, 3 * D, 4 * D, (1, 2, 3)),
          ('NN dX gru4', 0, 0, S, 4 * D, 3 * D, (1, 2)),
          ('NN self-loop', 0, 0, N, D, D, (1,))]


def main():
    dev = torch.device('cuda:0')
    for name, ta, tb, m, n, k, splits in SHAPES:
        a = torch.randn((k, m) if ta else (m, k), device=dev)
        b = torch.randn((n, k) if tb else (k, n), device=dev)
        out = torch.empty(m, n, device=dev)
        line = '%-13s M=%6d N=%6d K=%6d auto=%-3d |' % (name, m, n, k, K.auto_split_k(m, n, k))
        for sk in splits:
            for _ in range(3):
                K.gemm(a, b, ta=bool(ta), tb=bool(tb), out=out, split_k=sk)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                K.gemm(a, b, ta=bool(ta), tb=bool(tb), out=out, split_k=sk)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 20
            line += ' s%d:%.0fus/%.0fTF' % (sk, us, 2.0 * m * n * k / us / 1e6)
        print(line)


if __name__ == '__main__':
    main()
