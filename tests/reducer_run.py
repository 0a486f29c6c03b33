"""Child process of tests/test_gpu_step_plan.py: a few training steps inside a ONE-rank RCCL group with the reducer forced on
(RENET_FORCE_REDUCER=1: every bucket really goes through RCCL on the reducer's stream), or without a process group.
Prints one JSON line: losses, a checksum of the parameters, what the reducer did."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 're-net_amd'))


def main():
    forced = os.environ.get('RENET_FORCE_REDUCER') == '1'
    import torch.distributed as dist
    if forced:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', sys.argv[1])
        dist.init_process_group('nccl', rank=0, world_size=1)
    import model as M
    import ops
    import parallel
    import preprocess as P
    import synth
    dev = torch.device('cuda:0')
    quads, num_ent, num_rels, _ = synth.make_stream('ICEWS18', seed=5, num_t=40)
    gd = P.build_graph_dict(quads, num_rels)
    hs, ho = P.HistoryIndex(quads, 's', 10), P.HistoryIndex(quads, 'o', 10)
    torch.manual_seed(7)
    ops.reset_seed_counter()
    net = M.RENet(num_ent, 200, num_rels, dropout=0.5, seq_len=10, num_k=10)
    gen = torch.Generator().manual_seed(3)
    net.global_emb = {int(t): torch.randn(1, 1, 200, generator=gen) * 0.1 for t in gd}
    net.to(dev).train()
    opt = parallel.HipAdam(net, lr=1e-3, weight_decay=1e-5, max_norm=1.0)
    red = opt.reducer
    rng = np.random.RandomState(1)
    losses, early, ready, norms, used = [], [], [], [], []
    for k in range(4):
        idx = rng.permutation(len(quads))[:256]
        prep = net.prepare_both(quads[idx], hs.take(idx), ho.take(idx), gd)
        with opt.step_scope(head_passes=1):
            loss = net.loss_prepared_both(prep)
            used.append(type(loss.grad_fn).__name__)
            loss.backward()
            early.append([bk.work is not None for bk in red.buckets])        # launched DURING backward
            opt.step()
            ready.append(bool(red.partials_ready))
        losses.append(loss.item())
        norms.append(float(opt.norm.item()))
    torch.cuda.synchronize()
    params = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).double()
    out = {'losses': losses, 'norms': norms, 'early': early, 'ready': ready, 'used': used,
           'regions': [list(r) for r in red.regions()], 'psum': float(params.sum()), 'pabs': float(params.abs().sum()),
           'pnorm': float(params.norm())}
    opt.close()
    if forced:
        dist.destroy_process_group()
    print('RESULT ' + json.dumps(out))


if __name__ == '__main__':
    main()
