"""Three training steps (merged pass, HipAdam, dropout 0.5, seeded masks) of a small model on the ICEWS18-shaped stream; prints ONE JSON
line {"mode", "losses", "digest"} (sha256 of the final parameters).  `python tests/mode_run.py [per-model mode]`: with an argument the
model carries `net.gemm_mode = <mode>` inside a process of whatever default; without, the process default (RENET_GEMM) is used.
tests/test_gpu_streams.py compares the two forms bit for bit (the exact-fp32 mode became a per-model mode in round 5)."""
import hashlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 're-net_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    import model as M
    import ops
    import parallel
    import preprocess as P
    import renet_hip as K
    import synth
    K.lib()
    mode = sys.argv[1] if len(sys.argv) > 1 else None
    dev = torch.device('cuda:0')
    quads, num_ent, num_rels, _ = synth.make_stream('ICEWS18', seed=5, num_t=40)
    gd = P.build_graph_dict(quads, num_rels)
    hs, ho = P.HistoryIndex(quads, 's', 10), P.HistoryIndex(quads, 'o', 10)
    perm = np.random.RandomState(1).permutation(len(quads))
    torch.manual_seed(7)
    net = M.RENet(num_ent, 200, num_rels, dropout=0.5, seq_len=10, num_k=10)
    gen = torch.Generator().manual_seed(3)
    net.global_emb = {int(t): torch.randn(1, 1, 200, generator=gen) * 0.1 for t in gd}
    net.to(dev).train()
    net.gemm_mode = mode
    opt = parallel.HipAdam(net, lr=1e-3, weight_decay=1e-5, max_norm=1.0)
    losses = []
    for k in range(3):
        idx = perm[k * 512:(k + 1) * 512]
        ops.reset_seed_counter(1000 + 100 * k)
        with opt.step_scope(head_passes=1):
            loss = net.loss_prepared_both(net.prepare_both(quads[idx], hs.take(idx), ho.take(idx), gd))
            loss.backward()
            opt.step()
        losses.append(loss.item())
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).cpu().numpy()
    opt.close()
    print(json.dumps({'mode': mode or K.GEMM_MODE, 'process_default': K.GEMM_MODE, 'losses': losses,
                      'digest': hashlib.sha256(flat.tobytes()).hexdigest()}))


if __name__ == '__main__':
    main()
