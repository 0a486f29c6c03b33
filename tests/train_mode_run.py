"""TRAIN-MODE accuracy run of the HIP path (dropout 0.5 at all four sites), the counterpart of
tools/make_e2e_drop_golden.py (which runs the unmodified reference on CPU):

    per seed:  pretrain.py's loop for the global model (pretrain.py:60-96; 2 epochs)  ->  get_global_emb
               ->  train.py's loop (train.py:127-143) as the PRODUCT runs it (merged pass of both directions,
               fused clip + Adam)  ->  train.py's filtered validation (train.py:151-185)  ->  MRR / Hits@1,3,10

    python tests/train_mode_run.py [seed ...]      RENET_GEMM selects the GEMM mode (bf16x6 | f16x3 | f32 | bf16s)

Prints ONE JSON line {"gemm_mode", "seeds", "mrr": [...], "hits": [[h1,h3,h10],...], "epoch_loss": [[...],...], "seconds"}.
Used by tests/test_gpu_e2e.py (statistical comparison with tests/golden/e2e_yago_drop.npz) and by the GPU sessions that
write profiles/r04_train_mode_mrr.md.  Runs on cuda:0 only (the product has no CPU path)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 're-net_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def mrr_hits(ranks):
    ranks = np.asarray(ranks, dtype=np.float64).reshape(-1)
    return float(np.mean(1.0 / ranks)), [float(np.mean(ranks <= k)) for k in (1, 3, 10)]


FULL_CFG = dict(h=200, seq_len=10, batch=1024, num_k=1000, lr=1e-3, wd=1e-5, grad_norm=1.0, maxpool=1, pre_batch=1024,
                pre_lr=1e-2)       # tools/make_e2e_full_golden.py (train.py:211-236, pretrain.py:113-135)


def run_seed(seed, data, gold, stream=False, keep_ranks=False, log=None, samples=None, test_too=False):
    """gold: mapping with h / seq_len / batch / num_k / dropout / lr / wd / grad_norm / maxpool / pre_* / epochs.
    stream: validate with evaluate_filter_stream (one batch per timestamp; same ranks as the per-quadruple calls,
    tests/test_gpu_parity.py::test_evaluate_filter_stream_equals_sequential_calls) instead of train.py's loop.
    samples: [n_draws, num_k] entity samples recorded from the reference run of this seed (replayed in order by the
    validation advance: the CPU and GPU generators differ); None = draw on the device.
    test_too: after the validation pass continue as test.py does from train.py's checkpoint (the state AFTER validation,
    train.py:187-195; histories cut to seq_len, test.py:96-103) over the test split; the result gains (test_mrr, test_hits)."""
    from sklearn.utils import shuffle
    import global_model as GM
    import model as M
    import ops
    import parallel
    import preprocess as P
    import utils as U
    tr, va, te = data['train'], data['valid'], data['test']
    num_ent, num_rels = int(data['num_ent']), int(data['num_rels'])
    h, seq_len, batch, num_k = int(gold['h']), int(gold['seq_len']), int(gold['batch']), int(gold['num_k'])
    drop = float(gold['dropout'])
    dev = torch.device('cuda:0')
    allq = np.concatenate((tr, va, te))
    hs, ho = P.HistoryIndex(allq, 's', seq_len), P.HistoryIndex(allq, 'o', seq_len)
    rng_tr = np.arange(len(tr))
    rng_va = np.arange(len(tr), len(tr) + len(va))
    rng_te = np.arange(len(tr) + len(va), len(allq))
    sh, sht = hs.to_lists(rng_tr)
    oh, oht = ho.to_lists(rng_tr)
    graph_dict = U.build_graph_dict(tr, num_rels)
    times = np.unique(tr[:, 3])
    np.random.seed(seed)                                       # train.py:29-31
    torch.manual_seed(seed)
    ops.reset_seed_counter()
    gnet = GM.RENet_global(num_ent, h, num_rels, dropout=drop, model=0, seq_len=seq_len, num_k=num_k,
                           maxpool=int(gold['maxpool']))
    net = M.RENet(num_ent, h, num_rels, dropout=drop, model=0, seq_len=seq_len, num_k=num_k)
    gnet.to(dev)
    net.to(dev)
    # ---- pretrain.py:60-96 (torch's Adam: the global model is not on the timed path)
    gopt = torch.optim.Adam(gnet.parameters(), lr=float(gold['pre_lr']), weight_decay=1e-5)
    tp_s, tp_o = U.get_true_distribution(tr, num_ent)
    for ep in range(int(gold['pre_epochs'])):
        gnet.train()
        tt, ps, po = shuffle(times, tp_s, tp_o)
        for bt, bs, bo in U.make_batch(tt, ps, po, int(gold['pre_batch'])):
            loss = gnet(torch.from_numpy(bt), torch.from_numpy(bs), torch.from_numpy(bo), graph_dict)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(gnet.parameters(), float(gold['grad_norm']))
            gopt.step()
            gopt.zero_grad()
    gnet.eval()
    with torch.no_grad():
        gnet.global_emb = gnet.get_global_emb(times, graph_dict)
    net.global_emb = gnet.global_emb
    net.graph_dict = graph_dict
    # ---- train.py:118-143, product loop
    opt = parallel.HipAdam(net, lr=float(gold['lr']), weight_decay=float(gold['wd']), max_norm=float(gold['grad_norm']))
    epoch_losses = []
    for ep in range(int(gold['epochs'])):
        net.train()
        d_, a, b, c, d2 = shuffle(tr, sh, sht, oh, oht)
        tot = 0.0
        for bd, bs, bst, bo, bot in U.make_batch2(d_, a, b, c, d2, batch):
            prep = net.prepare_both(bd, (bs, bst), (bo, bot), graph_dict)
            loss = net.loss_prepared_both(prep)
            loss.backward()
            opt.step()
            tot += float(loss.item())
        epoch_losses.append(tot / (len(tr) / batch))
        if log:
            log('  seed %d epoch %d loss %.5f' % (seed, ep + 1, epoch_losses[-1]))
    # ---- train.py:151-185
    net.eval()
    total = torch.from_numpy(allq).to(dev)
    valid = torch.from_numpy(va)
    vs, vo, ts, to = hs.to_lists(rng_va), ho.to_lists(rng_va), hs.to_lists(rng_te), ho.to_lists(rng_te)
    ranks = []
    if samples is not None:
        pending = [torch.from_numpy(np.asarray(x, dtype=np.int64)).to(dev) for x in samples]
        net.sample_entities = lambda prob: pending.pop(0)
    with torch.no_grad():
        net.init_history(tr, (sh, sht), (oh, oht), valid, vs, vo, te, ts, to)
        net.latest_time = valid[0][3]
        if stream:
            ranks, _ = net.evaluate_filter_stream(valid, vs, vo, gnet, total)
        else:
            for i in range(len(va)):
                rk, _ = net.evaluate_filter(valid[i], (vs[0][i], vs[1][i]), (vo[0][i], vo[1][i]), gnet, total)
                ranks.append(rk)
        test_res = None
        if test_too:
            for ee in range(num_ent):                                   # test.py:96-103
                while len(net.s_hist_test[ee]) > seq_len:
                    net.s_hist_test[ee].pop(0)
                    net.s_hist_test_t[ee].pop(0)
                while len(net.o_hist_test[ee]) > seq_len:
                    net.o_hist_test[ee].pop(0)
                    net.o_hist_test_t[ee].pop(0)
            tranks, _ = net.evaluate_filter_stream(torch.from_numpy(te), ts, to, gnet, total)
            test_res = mrr_hits(tranks)
    opt.close()
    mrr, hits = mrr_hits(ranks)
    if keep_ranks:
        out = (mrr, hits, epoch_losses, np.asarray(ranks).reshape(len(va), -1))
        return out + (test_res,) if test_too else out
    return (mrr, hits, epoch_losses, test_res) if test_too else (mrr, hits, epoch_losses)


def main():
    import renet_hip as K
    K.lib()
    seeds = [int(x) for x in sys.argv[1:]] or [999, 1000, 1001]
    data = np.load(os.path.join(GOLDEN, 'yago_prefix_big.npz'))
    gold = np.load(os.path.join(GOLDEN, 'e2e_yago_drop.npz'))
    t0 = time.time()
    res = [run_seed(s, data, gold) for s in seeds]
    print(json.dumps({'gemm_mode': K.GEMM_MODE, 'seeds': seeds, 'mrr': [r[0] for r in res], 'hits': [r[1] for r in res],
                      'epoch_loss': [r[2] for r in res], 'seconds': time.time() - t0}))


if __name__ == '__main__':
    main()
