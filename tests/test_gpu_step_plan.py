"""The merged training step issued from C (csrc/step.cpp: renet_step_forward / renet_step_backward, step_plan.StepFn)
against the autograd path of ops.py on the same batches: BIT-identical losses, flat gradients and updated parameters -- the
C launch list passes every kernel the arguments the Python path passes (one iteration of the reference's train.py:136-142).
Also: the launch list really is what runs by default, the step counts its launches, and the host cost per step drops."""
import os
import sys
import time

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 're-net_amd'))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a HIP device')
    return torch.device('cuda:0')


def _stream(num_t=40, seed=5):
    import preprocess as P
    import synth
    quads, num_ent, num_rels, _ = synth.make_stream('ICEWS18', seed=seed, num_t=num_t)
    gd = P.build_graph_dict(quads, num_rels)
    return quads, num_ent, num_rels, gd, P.HistoryIndex(quads, 's', 10), P.HistoryIndex(quads, 'o', 10)


def _train(dev, data, plan, dropout, planes=True, steps=3, batch=256, hidden=200, side=True, device_builder=False):
    import model as M
    import ops
    import parallel
    import renet_hip as K
    import step_plan
    quads, num_ent, num_rels, gd, hs, ho = data
    old = (step_plan.ENABLED, K.PLANES, ops.SIDE_STREAM)
    step_plan.ENABLED, K.PLANES, ops.SIDE_STREAM = plan, planes, side
    try:
        torch.manual_seed(7)
        ops.reset_seed_counter()
        net = M.RENet(num_ent, hidden, num_rels, dropout=dropout, seq_len=10, num_k=10)
        gen = torch.Generator().manual_seed(3)
        net.global_emb = {int(t): torch.randn(1, 1, hidden, generator=gen) * 0.1 for t in gd}
        net.to(dev).train()
        opt = parallel.HipAdam(net, lr=1e-3, weight_decay=1e-5, max_norm=1.0)
        rng = np.random.RandomState(1)
        losses, flats = [], []
        used = []
        dstore = None
        if device_builder:
            import gpu_builder
            dstore = gpu_builder.DeviceStore(quads, hs, ho, gd, net.global_emb, num_ent, num_rels, dev)
        for k in range(steps):
            idx = rng.permutation(len(quads))[:batch]
            if dstore is not None:
                prep = None
                while prep is None:
                    prep = net.finish_prepare_device(net.prepare_both_device(idx, dstore))
            else:
                prep = net.prepare_both(quads[idx], hs.take(idx), ho.take(idx), gd)
            with opt.step_scope(head_passes=1):
                loss = net.loss_prepared_both(prep)
                used.append(type(loss.grad_fn).__name__)
                loss.backward()
                opt.sync_grads()
                flats.append(opt.grads.flat.detach().clone())
                opt.step()
            losses.append(loss.item())
        torch.cuda.synchronize()
        params = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()
        opt.close()
        return losses, flats, params, used
    finally:
        step_plan.ENABLED, K.PLANES, ops.SIDE_STREAM = old


@pytest.mark.parametrize('dropout', [0.0, 0.5])
@pytest.mark.parametrize('planes', [True, False])
def test_c_launch_list_is_bit_identical_to_the_autograd_path(dev, dropout, planes):
    data = _stream()
    la, fa, pa, ua = _train(dev, data, True, dropout, planes)
    lb, fb, pb, ub = _train(dev, data, False, dropout, planes)
    assert all('StepFn' in u for u in ua), ua                    # the C launch list really ran ...
    assert not any('StepFn' in u for u in ub), ub                # ... and the reference run really did not
    assert la == lb, (la, lb)
    for x, y in zip(fa, fb):
        assert float(x.abs().max()) > 0
        assert torch.equal(x, y)
    assert torch.equal(pa, pb)


def test_c_launch_list_on_one_stream_and_on_a_device_built_batch(dev):
    """RENET_SIDE_STREAM=0 (every launch in stream order) and a batch from the device builder (gpu_builder.DeviceBatch
    carries the same arrays as graph.DeviceGraph): still bit-identical to the autograd path."""
    data = _stream()
    a = _train(dev, data, True, 0.5, side=False)
    b = _train(dev, data, False, 0.5, side=False)
    assert a[0] == b[0] and torch.equal(a[2], b[2])
    c = _train(dev, data, True, 0.5, device_builder=True)
    d = _train(dev, data, False, 0.5, device_builder=True)
    assert c[0] == d[0] and torch.equal(c[2], d[2])
    assert all('StepFn' in u for u in c[3])


def test_c_launch_list_n_hidden_400_and_100(dev):
    for hidden in (100, 400):
        data = _stream(num_t=24)
        a = _train(dev, data, True, 0.5, hidden=hidden, steps=2, batch=128)
        b = _train(dev, data, False, 0.5, hidden=hidden, steps=2, batch=128)
        assert a[0] == b[0] and torch.equal(a[2], b[2]), hidden


def test_c_launch_list_materialises_gradients_when_no_buffers_exist(dev):
    """train.py's own loop (torch.optim.Adam + zero_grad(): .grad is None at every step): the C launch list still runs and its
    gradients come back through autograd.  Same loss bit for bit; the gradients equal the autograd path's up to fp32
    summation order only -- a parameter with several uses (ent_embeds: score heads, sequence assembly, RGCN) is summed
    here in ONE buffer in launch order, there as separately materialised terms that autograd adds."""
    import model as M
    import ops
    import step_plan
    quads, num_ent, num_rels, gd, hs, ho = _stream(num_t=24)
    idx = np.random.RandomState(4).permutation(len(quads))[:128]
    res = []
    for plan in (True, False):
        old = step_plan.ENABLED
        step_plan.ENABLED = plan
        try:
            torch.manual_seed(7)
            ops.reset_seed_counter()
            net = M.RENet(num_ent, 200, num_rels, dropout=0.5, seq_len=10, num_k=10)
            gen = torch.Generator().manual_seed(3)
            net.global_emb = {int(t): torch.randn(1, 1, 200, generator=gen) * 0.1 for t in gd}
            net.to(dev).train()
            prep = net.prepare_both(quads[idx], hs.take(idx), ho.take(idx), gd)
            assert prep is not None and step_plan.eligible(net, prep) == plan
            loss = net.loss_prepared_both(prep)
            assert ('StepFn' in type(loss.grad_fn).__name__) == plan
            (loss * 0.5).backward()
            torch.cuda.synchronize()
            res.append((loss.item(), {n: p.grad.clone() for n, p in net.named_parameters()}))
        finally:
            step_plan.ENABLED = old
    assert res[0][0] == res[1][0]
    for n in res[0][1]:
        a, b = res[0][1][n], res[1][1][n]
        assert float(b.abs().max()) > 0, n
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()), n


def test_c_launch_list_counts_its_launches_and_costs_less_host_time(dev):
    """Two C-ABI calls replace ~55: the launching thread's time per step (no synchronisation inside the loop) must drop
    well below the autograd path's; the step reports how many launches the C side issued."""
    import model as M
    import ops
    import parallel
    import step_plan
    quads, num_ent, num_rels, gd, hs, ho = _stream()
    net = M.RENet(num_ent, 200, num_rels, dropout=0.5, seq_len=10, num_k=10)
    gen = torch.Generator().manual_seed(3)
    net.global_emb = {int(t): torch.randn(1, 1, 200, generator=gen) * 0.1 for t in gd}
    net.to(dev).train()
    opt = parallel.HipAdam(net, lr=1e-3, weight_decay=1e-5, max_norm=1.0)
    rng = np.random.RandomState(2)
    preps = []
    for _ in range(12):
        idx = rng.permutation(len(quads))[:512]
        preps.append(net.prepare_both(quads[idx], hs.take(idx), ho.take(idx), gd))
    torch.cuda.synchronize()

    def run(plan):
        old = step_plan.ENABLED
        step_plan.ENABLED = plan
        try:
            for p in preps[:2]:
                with opt.step_scope(head_passes=1):
                    net.loss_prepared_both(p).backward()
                    opt.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for p in preps[2:]:
                with opt.step_scope(head_passes=1):
                    net.loss_prepared_both(p).backward()
                    opt.step()
            dt = (time.perf_counter() - t0) / len(preps[2:])
            torch.cuda.synchronize()
            return dt
        finally:
            step_plan.ENABLED = old
    t_py = run(False)
    t_c = run(True)
    fwd, bwd = step_plan.StepFn.last_launches
    assert 15 <= fwd <= 40 and 25 <= bwd <= 60, (fwd, bwd)
    assert t_c < 0.5 * t_py, (t_c, t_py)
    import copy
    twin = copy.deepcopy(net)                       # (the launch list's caches must not live inside the module)
    assert torch.equal(twin.ent_embeds, net.ent_embeds)
    opt.close()


def test_three_bucket_reducer_through_rccl_changes_nothing_but_the_norms_rounding(dev):
    """RENET_FORCE_REDUCER=1 in a one-rank RCCL group: the score head's and the encoders' buckets are all-reduced DURING the
    backward pass of the C launch list (events after their last gradient kernels), each region's sum of squares is taken
    behind its all-reduce, and the optimizer only combines them -- same losses, same gradient norm up to the order of
    the partial sums, same parameters up to that factor's rounding (parallel.OverlapReducer, train.py:140-142)."""
    import json
    import socket
    import subprocess
    res = {}
    for forced in ('1', '0'):
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, RENET_FORCE_REDUCER=forced, HSA_ENABLE_IPC_MODE_LEGACY='0')
        env.pop('RENET_REDUCER_BUCKETS', None)
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'reducer_run.py'), str(port)], env=env,
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        line = [l for l in out.stdout.splitlines() if l.startswith('RESULT ')][-1]
        res[forced] = json.loads(line[7:])
    a, b = res['1'], res['0']
    assert all('StepFn' in u for u in a['used'] + b['used'])
    assert len(a['regions']) == 3 and a['regions'][0][0] == 0                    # head | encoders | rest
    assert all(e == [True, True] for e in a['early']), a['early']               # both timed buckets left during backward
    assert all(a['ready']) and not any(b['ready'])
    assert all(e == [False, False] for e in b['early'])
    assert a['losses'][0] == b['losses'][0]
    for x, y in zip(a['losses'], b['losses']):
        assert abs(x - y) <= 2e-6 * abs(y), (a['losses'], b['losses'])
    for x, y in zip(a['norms'], b['norms']):
        assert x > 0 and abs(x - y) <= 2e-6 * y, (a['norms'], b['norms'])
    assert abs(a['pnorm'] - b['pnorm']) <= 1e-6 * b['pnorm'] and abs(a['pabs'] - b['pabs']) <= 1e-6 * b['pabs']
