#!/usr/bin/env python
"""Times renet_gemm_f32 on the GEMM shapes of one ICEWS18-shaped training step (GPU only)."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 're-net_amd'))
import renet_hip as K

N, S, B, E, D = 46075, 7624, 1024, 23033, 200
SHAPES = [
    ('NN self-loop', 0, 0, N, D, D), ('NN dX gru4', 0, 0, S, 4 * D, 3 * D), ('NN dX gru3', 0, 0, S, 3 * D, 3 * D),
    ('NN dfeat', 0, 0, B, 3 * D, E), ('NN gru-bwd step', 0, 0, 951, D, 3 * D),
    ('NT Gi gru4', 0, 1, S, 3 * D, 4 * D), ('NT Gi gru3', 0, 1, S, 3 * D, 3 * D), ('NT logits', 0, 1, B, E, 3 * D),
    ('NT dh loop', 0, 1, N, D, D), ('NT gru step', 0, 1, 951, 3 * D, D),
    ('TN dW_loop', 1, 0, D, D, N), ('TN dW_ih4', 1, 0, 3 * D, 4 * D, S), ('TN dW_hh', 1, 0, 3 * D, D, S),
    ('TN dW_lin', 1, 0, E, 3 * D, B), ('NN square 4096', 0, 0, 4096, 4096, 4096),
]


def main():
    dev = torch.device('cuda:0')
    for name, ta, tb, m, n, k in SHAPES:
        a = torch.randn((k, m) if ta else (m, k), device=dev)
        b = torch.randn((n, k) if tb else (k, n), device=dev)
        out = torch.empty(m, n, device=dev)
        ref = (a.t() if ta else a).double() @ (b.t() if tb else b).double() if m * n * k < 3e10 else None
        line = '%-16s M=%6d N=%6d K=%6d s=%-3d' % (name, m, n, k, K.auto_split_k(m, n, k))
        for mode in ('f32', 'bf16x6'):
            for _ in range(3):
                K.gemm(a, b, ta=bool(ta), tb=bool(tb), out=out, mode=mode)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            it = 20
            e0.record()
            for _ in range(it):
                K.gemm(a, b, ta=bool(ta), tb=bool(tb), out=out, mode=mode)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / it
            err = float((out.double() - ref).abs().max() / ref.abs().max()) if ref is not None else float('nan')
            line += ' | %-6s %8.1f us %6.1f TF err %.1e' % (mode, us, 2.0 * m * n * k / us / 1e6, err)
        print(line)


if __name__ == '__main__':
    main()
