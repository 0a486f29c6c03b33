"""End-to-end parity on REAL data (a prefix of YAGO): training from the same seed with train.py's loop
(seeds train.py:29-31, step :127-143) and filtered validation (train.py:151-185) on the HIP path vs the
same loop run with the UNMODIFIED reference modules on CPU (tools/make_e2e_golden.py -> tests/golden/).
North-star criterion: filtered MRR within +-0.002 of the reference."""
import os

import numpy as np
import pytest
import torch

from helpers import O, GOLDEN

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('tag,loop', [('', 'reference'), ('_big', 'reference'), ('', 'product')])
def test_yago_prefix_training_and_filtered_mrr_match_reference(tag, loop):
    """loop = 'reference': train.py's own step (two model() calls, clip_grad_norm_, torch Adam, zero_grad);
    loop = 'product': the path bench.py times -- merged pass of both directions (RENet.loss_prepared_both) and the
    fused clip + Adam + zero_grad on flat buffers (parallel.HipAdam) -- against the same reference trajectory."""
    from sklearn.utils import shuffle
    if not os.path.isfile(os.path.join(GOLDEN, 'e2e_yago%s.npz' % tag)):
        pytest.skip('fixture e2e_yago%s.npz not generated' % tag)
    import global_model as GM
    import model as M
    import preprocess as P
    import utils as U
    data = np.load(os.path.join(GOLDEN, 'yago_prefix%s.npz' % tag))
    gold = np.load(os.path.join(GOLDEN, 'e2e_yago%s.npz' % tag))
    tr, va, te = data['train'], data['valid'], data['test']
    num_ent, num_rels = int(data['num_ent']), int(data['num_rels'])
    h, seq_len, batch, num_k = int(gold['h']), int(gold['seq_len']), int(gold['batch']), int(gold['num_k'])
    dev = torch.device('cuda:0')
    allq = np.concatenate((tr, va, te))
    hs, ho = P.HistoryIndex(allq, 's', 10), P.HistoryIndex(allq, 'o', 10)
    rng_tr = np.arange(len(tr))
    rng_va = np.arange(len(tr), len(tr) + len(va))
    rng_te = np.arange(len(tr) + len(va), len(allq))
    sh, sht = hs.to_lists(rng_tr)
    oh, oht = ho.to_lists(rng_tr)
    graph_dict = U.build_graph_dict(tr, num_rels)
    np.random.seed(999)                                       # train.py:29-31
    torch.manual_seed(999)
    net = M.RENet(num_ent, h, num_rels, dropout=float(gold['dropout']), model=0, seq_len=seq_len, num_k=num_k)
    gnet = GM.RENet_global(num_ent, h, num_rels, dropout=float(gold['dropout']), model=0, seq_len=seq_len,
                           num_k=num_k, maxpool=int(gold['maxpool']))
    net.to(dev)
    gnet.to(dev)
    if loop == 'product':
        import parallel
        opt = parallel.HipAdam(net, lr=float(gold['lr']), weight_decay=float(gold['wd']),
                               max_norm=float(gold['grad_norm']))
    else:
        opt = torch.optim.Adam(net.parameters(), lr=float(gold['lr']), weight_decay=float(gold['wd']))
    with torch.no_grad():
        net.global_emb = gnet.get_global_emb(np.unique(tr[:, 3]), graph_dict)
    net.graph_dict = graph_dict
    step_losses, epoch_losses = [], []
    for ep in range(int(gold['epochs'])):
        net.train()
        d_, a, b, c, d2 = shuffle(tr, sh, sht, oh, oht)        # train.py:127 (same RNG stream as the reference run)
        tot = 0.0
        for bd, bs, bst, bo, bot in U.make_batch2(d_, a, b, c, d2, batch):
            prep = net.prepare_both(bd, (bs, bst), (bo, bot), graph_dict) if loop == 'product' else None
            bd = torch.from_numpy(bd).long().to(dev)
            if prep is not None:
                loss = net.loss_prepared_both(prep)
            else:
                loss = net(bd, (bs, bst), (bo, bot), graph_dict, subject=True) + \
                    net(bd, (bs, bst), (bo, bot), graph_dict, subject=False)
            loss.backward()
            if loop == 'product':
                opt.step()                                        # clip -> Adam -> zero_grad, one fused pass
            else:
                torch.nn.utils.clip_grad_norm_(net.parameters(), float(gold['grad_norm']))
                opt.step()
                opt.zero_grad()
            step_losses.append(loss.item())
            tot += step_losses[-1]
        epoch_losses.append(tot / (len(tr) / batch))
    step_losses = np.asarray(step_losses)
    # identical init + identical batches: the first steps must agree to fp32 rounding (measured 2e-6 over the
    # first 40 steps); afterwards the trajectories separate the way any two fp32 runs with different summation
    # orders do (Adam amplifies the last bit of every gradient).  Measured spread at step 164 / in the last epoch
    # mean, ours vs the reference: 6e-3 / 3.8e-3 with EXACT-fp32 products everywhere (RENET_GEMM=f32), 8e-3 /
    # 4.7e-3 with the bf16x6 kernels, 5e-3 / <2e-3 with an earlier bf16x6 build -- i.e. the spread between our own
    # exact and split variants is as large as their distance to the reference.  The bounds leave ~2x room; the
    # filtered MRR (the north-star criterion, +-0.002) is checked below.
    rel = np.abs(step_losses - gold['step_loss']) / gold['step_loss']
    erel = np.abs(np.asarray(epoch_losses) - gold['epoch_loss']) / gold['epoch_loss']
    print('drift vs reference: first step %.1e, max over steps <40 %.1e, max over all steps %.1e, epochs %s' % (
        rel[0], rel[:40].max(), rel.max(), np.array2string(erel, precision=1)))
    assert rel[0] < 1e-4
    assert rel[:40].max() < 1e-3 and rel.max() < 2e-2
    assert erel[0] < 1e-4 and erel.max() < 1e-2

    net.eval()
    gnet.eval()
    samples = [torch.from_numpy(x).to(dev) for x in gold['samples']]
    net.sample_entities = lambda prob: samples.pop(0)
    total = torch.from_numpy(allq).to(dev)
    valid = torch.from_numpy(va)
    vs, vo, ts, to = hs.to_lists(rng_va), ho.to_lists(rng_va), hs.to_lists(rng_te), ho.to_lists(rng_te)
    ranks = []
    with torch.no_grad():
        net.init_history(tr, (sh, sht), (oh, oht), valid, vs, vo, te, ts, to)
        net.latest_time = valid[0][3]
        for i in range(len(va)):
            rk, _ = net.evaluate_filter(valid[i], (vs[0][i], vs[1][i]), (vo[0][i], vo[1][i]), gnet, total)
            ranks.append(rk)
    ranks = np.asarray(ranks)
    mine, ref = O.mrr_hits(ranks.reshape(-1)), O.mrr_hits(gold['ranks'].reshape(-1))
    print('filtered MRR mine %.6f reference %.6f | hits@10 %.4f / %.4f' % (mine['mrr'], ref['mrr'], mine['hits@10'],
                                                                            ref['hits@10']))
    assert abs(mine['mrr'] - ref['mrr']) <= 0.002, (mine, ref)
    for k in ('hits@1', 'hits@3', 'hits@10'):
        assert abs(mine[k] - ref[k]) <= 0.01, (k, mine[k], ref[k])


# every mode whose rate bench.py prints has its accuracy test in the default suite: bf16x6 (`value`), f16x3 (`modes.value_f16x3`),
# bf16s (`other_configs.yago_d400_l15_bf16`).  Exact fp32 equals bf16x6 to 1e-6 per seed (profiles/r04_c_train_mode_mrr.md, r05_c
# section 1b): on request only (RENET_TEST_ALL_MODES=1).
_TRAIN_MODES = ['bf16x6', 'f16x3', 'bf16s'] + (['f32'] if os.environ.get('RENET_TEST_ALL_MODES') == '1' else [])


@pytest.mark.parametrize('mode', _TRAIN_MODES)
def test_train_mode_filtered_mrr_matches_reference_statistically(mode):
    """The mode bench.py TIMES -- dropout 0.5 at all four sites (RGCN.py:36-37, Aggregator.py:157-158, model.py:90,99) --
    against the unmodified reference trained the same way (tools/make_e2e_drop_golden.py -> tests/golden/e2e_yago_drop.npz:
    YAGO prefix, global model pretrained for 2 epochs, 4 epochs of train.py's loop, filtered validation, several seeds).
    Dropout masks differ by construction (torch's CPU generator vs the kernels' counters), so the comparison is of the
    per-seed MEANS:  |mean MRR(HIP) - mean MRR(reference)| <= 0.002 + 2 x pooled standard error  (the north-star
    tolerance plus what seeds alone move the mean by), once per GEMM mode: bf16x6 (default, 24-bit split), f16x3
    (22-bit split), f32 (exact products), bf16s (bf16 operand storage).  tests/train_mode_run.py runs the product loop
    in a child process (RENET_GEMM is read at import)."""
    import json
    import subprocess
    import sys
    gpath = os.path.join(GOLDEN, 'e2e_yago_drop.npz')
    if not os.path.isfile(gpath):
        pytest.skip('fixture e2e_yago_drop.npz not generated')
    gold = np.load(gpath)
    ref = np.asarray(gold['mrr'], dtype=np.float64)
    if len(ref) < 3:
        pytest.skip('fixture holds fewer than 3 reference seeds')
    seeds = [int(x) for x in gold['seeds']]            # every seed the fixture holds (4 since round 4)
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, 'train_mode_run.py')] + [str(x) for x in seeds],
                       env=dict(os.environ, RENET_GEMM=mode), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert out['gemm_mode'] == mode
    mine = np.asarray(out['mrr'], dtype=np.float64)
    ref = ref[:len(mine)]
    se = float(np.sqrt(ref.var(ddof=1) / len(ref) + mine.var(ddof=1) / len(mine)))
    diff = float(mine.mean() - ref.mean())
    print('train-mode filtered MRR [%s]: mine %s mean %.6f | reference %s mean %.6f | diff %+.6f, pooled s.e. %.6f '
          '(%.0f s)' % (mode, np.array2string(mine, precision=4), mine.mean(), np.array2string(ref, precision=4),
                        ref.mean(), diff, se, out['seconds']))
    assert abs(diff) <= 0.002 + 2.0 * se, (mode, mine.tolist(), ref.tolist())
    # PAIRED by seed (the sharper statement: a seed fixes initialisation and batch order on both sides, only the dropout
    # masks differ): observed per-seed differences on MI355X +0.0007 / -0.0018 / -0.0029 in every fp32-class mode
    d = mine - ref[:len(mine)]
    se_p = float(d.std(ddof=1) / np.sqrt(len(d)))
    print('   paired by seed: differences %s, mean %+.6f, s.e. %.6f' % (np.array2string(d, precision=4), d.mean(), se_p))
    assert abs(d.mean()) <= 0.002 + 2.0 * se_p, (mode, d.tolist())
    # the epoch losses (means over dropout noise as well) must agree much more tightly than the ranks
    el_ref = np.asarray(gold['epoch_loss'], dtype=np.float64)[:, -1]
    el = np.asarray(out['epoch_loss'], dtype=np.float64)[:, -1]
    assert abs(el.mean() - el_ref.mean()) <= 0.02 * el_ref.mean() + 2.0 * np.sqrt(el.var(ddof=1) / len(el) +
                                                                                el_ref.var(ddof=1) / len(el_ref))


def _full_run(dropout, epochs, pre_epochs, seeds, mode=None):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    if mode:
        env['RENET_GEMM'] = mode
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'yago_full_run.py'), str(dropout), str(epochs),
                        str(pre_epochs)] + [str(s) for s in seeds], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])


def test_full_yago_deterministic_training_matches_the_reference():
    """ALL of public YAGO (161 540 train / 19 523 valid quadruples), the reference's defaults (train.py:211-236: lr 1e-3,
    batch 1024, num_k 1000; pretrain.py:113-135 for the global model), dropout 0 so that both sides are deterministic given
    the seed: the UNMODIFIED reference trained on CPU (tools/make_e2e_full_golden.py -> tests/golden/e2e_yago_full_d0.npz)
    vs the product loop on the MI355X (tools/yago_full_run.py), the reference's recorded entity samples replayed in the
    validation advance.  North-star criterion, no statistical allowance: |filtered MRR difference| <= 0.002."""
    gpath = os.path.join(GOLDEN, 'e2e_yago_full_d0.npz')
    if not os.path.isfile(gpath):
        pytest.skip('fixture e2e_yago_full_d0.npz not generated')
    gold = np.load(gpath)
    out = _full_run(0.0, int(gold['epochs']), int(gold['pre_epochs']), [int(gold['seeds'][0])])
    run = out['runs'][0]
    assert run['replayed_reference_samples']
    ref_mrr, ref_hits = float(gold['mrr'][0]), [float(x) for x in gold['hits'][0]]
    el, el_ref = np.asarray(run['epoch_loss']), np.asarray(gold['epoch_loss'][0], dtype=np.float64)
    print('full YAGO, dropout 0, %d epochs: filtered MRR mine %.6f reference %.6f | hits@1/3/10 %s / %s | epoch loss %s / %s | '
          'ranks equal %.4f (%.0f s)' % (int(gold['epochs']), run['mrr'], ref_mrr, np.round(run['hits'], 4), np.round(ref_hits, 4),
                                        np.round(el, 5), np.round(el_ref, 5), run['ranks_equal_frac'], run['seconds']))
    assert abs(run['mrr'] - ref_mrr) <= 0.002, (run['mrr'], ref_mrr)
    for a, b in zip(run['hits'], ref_hits):
        assert abs(a - b) <= 0.005, (run['hits'], ref_hits)
    assert abs(el[0] - el_ref[0]) <= 1e-3 * el_ref[0]              # first epoch: 158 steps from identical parameters
    assert np.all(np.abs(el - el_ref) <= 1e-2 * el_ref), (el, el_ref)


def test_full_yago_train_mode_mrr_matches_the_reference_paired_by_seed():
    """The same at the reference's DEFAULT dropout 0.5 (the mode bench.py times), >= 3 seeds x 3 epochs over all of YAGO
    (tests/golden/e2e_yago_full_drop.npz).  A seed fixes initialisation, batch order and -- replayed from the fixture -- the
    entity samples of the validation advance; only the dropout masks differ (torch's CPU generator vs the kernels' counters).
    Criterion: |mean over seeds of the PAIRED MRR differences| <= 0.002 -- the north star's tolerance, with NO statistical
    allowance (review r5: an "or within 2 s.e." clause would have admitted 0.003 at the measured s.e. of 0.0015; the measured
    mean is -0.0002).  The standard error is printed for the record."""
    gpath = os.path.join(GOLDEN, 'e2e_yago_full_drop.npz')
    if not os.path.isfile(gpath):
        pytest.skip('fixture e2e_yago_full_drop.npz not generated')
    gold = np.load(gpath)
    seeds = [int(s) for s in gold['seeds']]
    if len(seeds) < 3:
        pytest.skip('fixture holds fewer than 3 reference seeds')
    out = _full_run(float(gold['dropout']), int(gold['epochs']), int(gold['pre_epochs']), seeds)
    mine = np.asarray([r['mrr'] for r in out['runs']], dtype=np.float64)
    ref = np.asarray(gold['mrr'], dtype=np.float64)[:len(mine)]
    d = mine - ref
    se = float(d.std(ddof=1) / np.sqrt(len(d)))
    print('full YAGO, dropout %.1f, %d epochs, seeds %s: filtered MRR mine %s | reference %s | paired differences %s, mean '
          '%+.6f, s.e. %.6f (%.0f s)' % (float(gold['dropout']), int(gold['epochs']), seeds, np.round(mine, 5), np.round(ref, 5),
                                        np.round(d, 5), d.mean(), se, out['seconds']))
    assert abs(d.mean()) <= 0.002, (mine.tolist(), ref.tolist())
    el = np.asarray([r['epoch_loss'][-1] for r in out['runs']])
    el_ref = np.asarray(gold['epoch_loss'], dtype=np.float64)[:len(el), -1]
    assert abs(el.mean() - el_ref.mean()) <= 0.01 * el_ref.mean(), (el, el_ref)


@pytest.mark.skipif(os.environ.get('RENET_TEST_FULL_LENGTH') == '0',
                    reason='RENET_TEST_FULL_LENGTH=0: the 20 + 20-epoch run (~70 s of GPU) was switched off')
def test_full_length_yago_run_lands_where_the_reference_lands():
    """The README schedule (global model 20 epochs at lr 1e-3, RE-Net 20 epochs, dropout 0.5) on all of YAGO, seed 999, validation
    and test split as train.py / test.py run them, against the UNMODIFIED reference trained the same way on CPU for 5 hours
    (tools/make_full_length_reference.py -> tests/golden/e2e_yago_full_len20.npz: test MRR 0.6479).  The masks differ, so the bound
    is the HIP seed spread measured over five seeds (sd 0.0022, profiles/r05_c section 3): |test MRR difference| <= 0.006."""
    gpath = os.path.join(GOLDEN, 'e2e_yago_full_len20.npz')
    if not os.path.isfile(gpath):
        pytest.skip('fixture e2e_yago_full_len20.npz not generated')
    import json
    import subprocess
    import sys
    gold = np.load(gpath)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RENET_FULL_TEST='1', RENET_FULL_PRE_LR=str(float(gold['pre_lr'])))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'yago_full_run.py'), str(float(gold['dropout'])),
                        str(int(gold['epochs'])), str(int(gold['pre_epochs'])), str(int(gold['seed']))], env=env,
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    run = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])['runs'][0]
    el, el_ref = np.asarray(run['epoch_loss']), np.asarray(gold['epoch_loss'], dtype=np.float64)
    print('full-length YAGO (20 + 20 epochs), seed %d: TEST MRR mine %.4f reference %.4f | hits %s / %s | valid MRR %.4f / %.4f | '
          'epoch-20 loss %.4f / %.4f (%.0f s)' % (int(gold['seed']), run['test_mrr'], float(gold['test_mrr']),
                                                 np.round(run['test_hits'], 4), np.round(gold['test_hits'], 4), run['mrr'],
                                                 float(gold['valid_mrr']), el[-1], el_ref[-1], run['seconds']))
    assert abs(run['test_mrr'] - float(gold['test_mrr'])) <= 0.006
    assert abs(run['mrr'] - float(gold['valid_mrr'])) <= 0.008
    assert np.all(np.abs(el - el_ref) <= 0.015 * el_ref), (el, el_ref)
