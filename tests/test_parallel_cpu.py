"""World-size-2 `gloo` test (CPU) of the data-parallel exchange: flat gradient views + one all-reduce
== single-process gradient accumulation over the same two batches (re-net_amd/parallel.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import parallel


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(5)
    return torch.nn.Sequential(torch.nn.Linear(12, 16), torch.nn.Tanh(), torch.nn.Linear(16, 7))


def _data(step, rank, world):
    perm = np.random.RandomState(3).permutation(400)
    idx = parallel.shard_indices(perm, step, rank, world, 20)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(400, 12, generator=g)
    y = torch.randint(0, 7, (400,), generator=g)
    return x[idx], y[idx], idx


def _flat(net):
    """gradients in parallel.flat_layout order (every tensor padded to a multiple of 4 floats)"""
    parts = []
    for p in net.parameters():
        g = p.grad.reshape(-1)
        parts.append(torch.cat((g, torch.zeros((-g.numel()) % 4))))
    return torch.cat(parts)


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    net = _model()
    flat = parallel.FlatGrads(net)
    res = []
    for step in range(2):
        x, y, _ = _data(step, rank, world)
        torch.nn.functional.cross_entropy(net(x), y).backward()
        assert flat.check_views()
        flat.allreduce_mean()
        norm = flat.clip_(0.05)
        res.append((flat.flat.clone(), float(norm)))
        flat.zero()
    if rank == 0:
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


def test_flat_allreduce_equals_accumulation(tmp_path):
    world, port, out = 2, _free_port(), str(tmp_path / 'r0.pt')
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    got = torch.load(out)
    net = _model()
    for step in range(2):
        grads = []
        seen = []
        for rank in range(world):
            net.zero_grad()
            x, y, idx = _data(step, rank, world)
            seen.append(set(idx.tolist()))
            torch.nn.functional.cross_entropy(net(x), y).backward()
            grads.append(_flat(net))
        assert not (seen[0] & seen[1]), 'ranks must see disjoint slices of the shuffled order'
        ref = (grads[0] + grads[1]) / world
        norm = ref.norm()
        ref = ref * torch.clamp(0.05 / (norm + 1e-6), max=1.0)
        assert abs(float(norm) - got[step][1]) < 1e-6
        torch.testing.assert_close(got[step][0], ref, rtol=1e-6, atol=1e-7)


def test_single_process_paths_are_noops():
    net = _model()
    flat = parallel.FlatGrads(net)
    x, y, _ = _data(0, 0, 1)
    torch.nn.functional.cross_entropy(net(x), y).backward()
    before = flat.flat.clone()
    flat.allreduce_mean()                      # no process group: must not touch the gradient
    assert torch.equal(before, flat.flat) and flat.check_views()
    assert torch.equal(_flat(net), flat.flat)


# ---------------------------------------------------------------------------------------------
# RENet-level: two ranks, each with its own batch, real model + real flat-gradient layout + the overlapped
# two-bucket exchange == one process accumulating the same two batches.  The container has no GPU: the device
# wrappers are emulated in torch-CPU (tests/cpu_abi_emulation.py); the exchange logic under test is the product's.
# ---------------------------------------------------------------------------------------------
def _renet_setup():
    import model as M
    import preprocess as P
    import synth
    quads, num_ent, num_rels, _ = synth.make_stream('YAGO', seed=7, num_t=30)
    quads = quads[quads[:, 0] < 400]
    quads = quads[quads[:, 2] < 400]
    num_ent = 400
    gd = P.build_graph_dict(quads, num_rels)
    hs, ho = P.HistoryIndex(quads, 's'), P.HistoryIndex(quads, 'o')
    torch.manual_seed(21)
    net = M.RENet(num_ent, 100, num_rels, dropout=0.0, seq_len=10)
    gen = torch.Generator().manual_seed(2)
    net.global_emb = {int(t): torch.randn(1, 1, 100, generator=gen) * 0.1 for t in gd}
    net.eval()
    perm = np.random.RandomState(4).permutation(len(quads))
    return net, quads, gd, hs, ho, perm


def _renet_grads(net, quads, gd, hs, ho, idx, pair):
    b = quads[idx]
    ps = net.prepare(b, hs.take(idx), gd, subject=True)
    po = net.prepare(b, ho.take(idx), gd, subject=False)
    loss = net.loss_prepared_pair(ps, po) if pair else net.loss_prepared(ps) + net.loss_prepared(po)
    loss.backward()
    return float(loss)


def _renet_worker(rank, world, port, out):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import cpu_abi_emulation
    cpu_abi_emulation.install()
    import ops
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    net, quads, gd, hs, ho, perm = _renet_setup()
    flat = parallel.FlatGrads(net)
    early = [p for n, p in net.named_parameters() if n in ('linear.weight', 'linear.bias')]
    red = parallel.OverlapReducer(flat, flat.span(('linear.weight', 'linear.bias'), net), early)
    started = []
    orig = red.on_grad_done

    def spy(p):
        orig(p)
        started.append(red.work is not None)
    ops.register_grad_done_hook(early, spy)
    res = []
    # a backward pass OUTSIDE a declared step (smoke / eval with grad / exception before step()) must
    # never launch the early bucket, and finish() must still produce the exchanged gradient (synchronous path)
    idx = parallel.shard_indices(perm, 5, rank, world, 96)
    _renet_grads(net, quads, gd, hs, ho, idx, pair=False)
    assert started and not any(started), started
    red.finish()
    undeclared = flat.flat.clone()
    flat.zero()
    del started[:]
    for step in range(2):
        idx = parallel.shard_indices(perm, step, rank, world, 96)
        red.begin_step(head_passes=2)
        _renet_grads(net, quads, gd, hs, ho, idx, pair=(step == 1))
        assert flat.check_views()
        assert started[-1] and not any(started[:-1][-3:]), started      # launched by the LAST of the 4 notifications
        red.finish()
        res.append(flat.flat.clone())
        flat.zero()
        del started[:]
    # declaring too few head passes: the second pass would accumulate into a bucket that is being reduced -> error
    red.begin_step(head_passes=1)
    idx = parallel.shard_indices(perm, 0, rank, world, 96)
    try:
        _renet_grads(net, quads, gd, hs, ho, idx, pair=False)
        raised = False
    except RuntimeError as e:
        raised = 'declared too few passes' in str(e)
    red.begin_step(head_passes=2)          # re-arming completes the abandoned collective on every rank
    red.armed = False
    flat.zero()
    seeds = [ops.next_seed() for _ in range(3)]
    torch.save({'flat': res, 'seeds': seeds, 'undeclared': undeclared, 'raised': raised}, out % rank)
    dist.barrier()
    dist.destroy_process_group()


def test_renet_two_ranks_equal_accumulation_over_the_same_batches(tmp_path):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    world, port, out = 2, _free_port(), str(tmp_path / 'r%d.pt')
    mp.spawn(_renet_worker, args=(world, port, out), nprocs=world, join=True)
    got = [torch.load(out % r) for r in range(world)]
    assert torch.equal(got[0]['flat'][0], got[1]['flat'][0])          # both ranks hold the same averaged gradient
    assert torch.equal(got[0]['undeclared'], got[1]['undeclared']) and float(got[0]['undeclared'].abs().max()) > 0
    assert got[0]['raised'] and got[1]['raised']
    assert set(got[0]['seeds']).isdisjoint(got[1]['seeds'])           # dropout seeds differ per rank
    import cpu_abi_emulation
    undo = cpu_abi_emulation.install()
    try:
        net, quads, gd, hs, ho, perm = _renet_setup()
        flat = parallel.FlatGrads(net)
        for step in range(2):
            acc = torch.zeros_like(flat.flat)
            for rank in range(world):
                flat.zero()
                _renet_grads(net, quads, gd, hs, ho, parallel.shard_indices(perm, step, rank, world, 96), pair=False)
                acc += flat.flat
            ref = acc / world
            scale = float(ref.abs().max())
            assert float((got[0]['flat'][step] - ref).abs().max()) <= 1e-5 * scale
    finally:
        undo()


# ---------------------------------------------------------------------------------------------
# Round 6: the THREE-bucket exchange (score head | encoders | rest) against the two-bucket one on the same steps, and the
# per-region sums of squares the optimizer combines into the clip norm (train.py:140).
# ---------------------------------------------------------------------------------------------
def _three_bucket_worker(rank, world, port, out):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import cpu_abi_emulation
    cpu_abi_emulation.install()
    import ops
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    net, quads, gd, hs, ho, perm = _renet_setup()
    named = list(net.named_parameters())
    head = [p for n, p in named if n in ('linear.weight', 'linear.bias')]
    mid = [p for n, p in named if n.startswith(('encoder.', 'encoder_r.'))]
    mid_names = tuple(n for n, p in named if n.startswith(('encoder.', 'encoder_r.')))
    flat = parallel.FlatGrads(net, first=head + mid)                 # HipAdam's layout: head, encoders, rest
    res = {}
    for buckets in (2, 3):
        red = parallel.OverlapReducer(flat, flat.span(('linear.weight', 'linear.bias'), net), head,
                                      mid_span=flat.span(mid_names, net) if buckets == 3 else None,
                                      mid_params=mid if buckets == 3 else ())
        log = []
        orig = red.on_grad_done

        def spy(p, red=red, orig=orig, log=log):
            orig(p)
            log.append(tuple(bk.work is not None for bk in red.buckets))
        token = ops.register_grad_done_hook(head + mid, spy)
        flats, norms, launched = [], [], []
        for step, (pair, passes) in enumerate(((False, 2), (True, 2))):
            idx = parallel.shard_indices(perm, step, rank, world, 96)
            red.begin_step(head_passes=passes)
            _renet_grads(net, quads, gd, hs, ho, idx, pair=pair)
            launched.append(tuple(bk.work is not None for bk in red.buckets))     # BEFORE finish(): the early launches
            red.finish()
            assert red.partials_ready and len(red.bucket_sumsq) == len(red.regions())
            flats.append(flat.flat.clone())
            norms.append((red.total_norm(), float(torch.linalg.vector_norm(flat.flat.double()))))
            flat.zero()
        ops.unregister_grad_done_hook(token)
        res[buckets] = {'flat': flats, 'norms': norms, 'launched': launched, 'regions': red.regions(), 'log': log}
    torch.save(res, out % rank)
    dist.barrier()
    dist.destroy_process_group()


def test_three_bucket_exchange_equals_the_two_bucket_one(tmp_path):
    world, port, out = 2, _free_port(), str(tmp_path / 'b%d.pt')
    mp.spawn(_three_bucket_worker, args=(world, port, out), nprocs=world, join=True)
    got = [torch.load(out % r) for r in range(world)]
    for r in range(world):
        two, three = got[r][2], got[r][3]
        assert len(three['regions']) == 3, three['regions']
        assert three['regions'][0][0] == 0 and three['regions'][1][0] == three['regions'][0][1]      # head | encoders | rest
        for a, b in zip(two['flat'], three['flat']):
            assert float(a.abs().max()) > 0 and torch.equal(a, b)      # the SAME exchanged gradient, bit for bit
        # both timed buckets were in flight before finish(): the encoders' all-reduce overlaps the rest of the backward pass
        assert all(l == (True,) for l in two['launched']), two['launched']
        assert all(l == (True, True) for l in three['launched']), three['launched']
        for tn, ref in three['norms'] + two['norms']:
            assert abs(tn - ref) <= 1e-9 * ref                       # sqrt(sum of the per-region sums of squares) = ||g||
    assert torch.equal(got[0][3]['flat'][0], got[1][3]['flat'][0])     # and both ranks hold it


# ---------------------------------------------------------------------------------------------
# SURVEY 8e option (i): the EXACT split -- both ranks build the SAME reference batch, keep half of its sequences
# (graph.shard_sequences) and SUM their gradients: == the single-process step on that batch.
# ---------------------------------------------------------------------------------------------
def _renet_exact_worker(rank, world, port, out):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import cpu_abi_emulation
    cpu_abi_emulation.install()
    import ops
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    net, quads, gd, hs, ho, perm = _renet_setup()
    flat = parallel.FlatGrads(net)
    early = [p for n, p in net.named_parameters() if n in ('linear.weight', 'linear.bias')]
    red = parallel.OverlapReducer(flat, flat.span(('linear.weight', 'linear.bias'), net), early)
    ops.register_grad_done_hook(early, red.on_grad_done)
    red.begin_step(head_passes=1, average=False)
    idx = perm[:120]
    prep = net.prepare_both(quads[idx], hs.take(idx), ho.take(idx), gd, shard=(rank, world))
    loss = net.loss_prepared_both(prep)
    loss.backward()
    red.finish()
    lsum = torch.tensor([float(loss)])
    dist.all_reduce(lsum)
    torch.save({'flat': flat.flat.clone(), 'loss': float(lsum), 'rows': prep.b}, out % rank)
    dist.barrier()
    dist.destroy_process_group()


def test_exact_split_of_one_batch_over_two_ranks(tmp_path):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    world, port, out = 2, _free_port(), str(tmp_path / 'x%d.pt')
    mp.spawn(_renet_exact_worker, args=(world, port, out), nprocs=world, join=True)
    got = [torch.load(out % r) for r in range(world)]
    assert torch.equal(got[0]['flat'], got[1]['flat'])
    import cpu_abi_emulation
    undo = cpu_abi_emulation.install()
    try:
        net, quads, gd, hs, ho, perm = _renet_setup()
        flat = parallel.FlatGrads(net)
        idx = perm[:120]
        full = net.prepare_both(quads[idx], hs.take(idx), ho.take(idx), gd)
        assert got[0]['rows'] + got[1]['rows'] == full.b
        loss = net.loss_prepared_both(full)
        loss.backward()
        assert abs(got[0]['loss'] - float(loss)) <= 2e-6 * abs(float(loss))
        scale = float(flat.flat.abs().max())
        assert float((got[0]['flat'] - flat.flat).abs().max()) <= 1e-5 * scale
    finally:
        undo()


# ---------------------------------------------------------------------------------------------
# 8 ranks on one node, each with its own builder workers (pipeline.BatchPrefetcher): the per-rank worker budget
# must keep the node within its cores ("8 ranks x 16 prefetch workers do not oversubscribe"), and the
# prefetchers of all ranks must deliver their batches in step order while running side by side.
# ---------------------------------------------------------------------------------------------
def _prefetch_worker(rank, world, port, out):
    import pipeline
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    cpus = 256                                  # an 8-GPU MI355X node's host
    w = pipeline.worker_budget(world, cpus=cpus)
    tot = torch.tensor([w + 1])                 # workers + the training thread of this rank
    dist.all_reduce(tot)
    here = pipeline.worker_budget(world)        # this container (8 cores): 1 worker per rank => built inline
    pf = pipeline.BatchPrefetcher(lambda step: (rank, step, int(np.arange(step + 1).sum())), range(10, 22),
                                  max(here, 2) if rank == 0 else here)     # rank 0 really forks two workers
    got = list(pf)
    torch.save({'total': int(tot), 'cpus': cpus, 'workers': w, 'here': here, 'got': got}, out % rank)
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_with_prefetch_workers_stay_within_the_cores(tmp_path):
    import pipeline
    world, port, out = 8, _free_port(), str(tmp_path / 'p%d.pt')
    mp.spawn(_prefetch_worker, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        g = torch.load(out % r)
        assert g['workers'] == 16 and g['total'] <= g['cpus'] // 1 and g['total'] == world * 17
        assert g['got'] == [(r, s, s * (s + 1) // 2) for s in range(10, 22)]
    for cpus in (8, 32, 96, 128, 256):
        for w in (1, 2, 4, 8):
            n = pipeline.worker_budget(w, cpus=cpus)
            assert n >= 1 and (n == 1 or w * (n + 1) <= cpus)


def test_exact_split_shares_graph_site_dropout_seeds_across_ranks(monkeypatch):
    """in the exact split every rank evaluates the SAME batch graph -- its RGCN dropout masks must be the
    same on every rank (rank-independent seeds for graph sites), while the per-sequence sites keep rank-dependent
    seeds; without the switch every site is rank dependent."""
    import ops

    def seeds(rank, shared):
        monkeypatch.setattr(ops, '_rank', lambda: rank)
        monkeypatch.setattr(ops, 'SHARED_GRAPH_SEEDS', shared)
        ops.reset_seed_counter(0)
        return [ops.next_seed(graph_site=True), ops.next_seed(graph_site=True), ops.next_seed(), ops.next_seed()]
    a, b = seeds(0, True), seeds(3, True)
    assert a[:2] == b[:2] and a[2] != b[2] and a[3] != b[3]
    c, d = seeds(0, False), seeds(3, False)
    assert all(x != y for x, y in zip(c, d))
    assert len(set(a)) == 4
    ops.reset_seed_counter(0)


def test_bench_starts_its_own_ranks_when_no_launcher_is_around():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (what a driver does that does not wrap the
    command in torch.distributed.run): the script re-executes itself under torch.distributed.run with 2 ranks on
    127.0.0.1; RENET_BENCH_LAUNCH_CHECK=1 makes the ranks rendezvous over gloo and print the launch fields instead of
    measuring (no GPU here).  Under a launcher (WORLD_SIZE set) it must NOT spawn again."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR',
                                                             'MASTER_PORT')}
    env['RENET_BENCH_LAUNCH_CHECK'] = '1'
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1'], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert line == {'launcher_check': True, 'n_gpus': 2, 'rccl_ranks_seen': 2, 'gpus_arg': 2}
    # a rank of an existing launch: no re-spawn (one process, reports its own world)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2'],
                       env=dict(env, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0'), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 1 and line['gpus_arg'] == 2


# ---------------------------------------------------------------------------------------------
# parallel.ScalingMode -- the bookkeeping bench.py's three scaling modes share (which quadruples a rank takes, the global
# batch a line reports, whether gradients are averaged or summed) -- and the three modes END TO END on two gloo ranks over
# the emulated wrappers: weak / strong == one process accumulating the ranks' batches and averaging; exact == the 1-rank step.
# ---------------------------------------------------------------------------------------------
def test_scaling_mode_bookkeeping():
    perm = np.random.RandomState(0).permutation(100000)
    for world in (1, 2, 4, 8):
        weak, strong, exact = (parallel.ScalingMode(s, 1024, world) for s in ('weak', 'strong', 'exact'))
        assert (weak.rank_batch, weak.global_batch, weak.average, weak.reported_scaling) == (1024, 1024 * world, True, 'weak')
        assert (strong.rank_batch, strong.global_batch, strong.average) == (1024 // world, 1024, True)
        assert (exact.rank_batch, exact.global_batch, exact.average) == (1024, 1024, False)
        assert strong.reported_scaling == exact.reported_scaling == 'strong'
        assert exact.passes == 'merged' and parallel.ScalingMode('weak', 1024, world, passes='pair').passes == 'pair'
        for step in (0, 3):
            got = {m.scaling: [m.indices(perm, step, r) for r in range(world)] for m in (weak, strong, exact)}
            for name, m in (('weak', weak), ('strong', strong)):
                allq = np.concatenate(got[name])
                assert len(allq) == m.global_batch == len(np.unique(allq))          # disjoint shards, global batch covered
            for r in range(world):
                assert np.array_equal(got['exact'][r], got['exact'][0]) and len(got['exact'][r]) == 1024
                assert exact.shard(r) == ((r, world) if world > 1 else None)
            assert weak.shard(0) is None and strong.shard(0) is None
        # consecutive steps of the weak mode never hand two ranks the same slice
        a = np.concatenate([weak.indices(perm, 0, r) for r in range(world)] + [weak.indices(perm, 1, r) for r in range(world)])
        assert len(np.unique(a)) == len(a)
    with pytest.raises(ValueError):
        parallel.ScalingMode('linear', 1024, 2)


def _mode_step(net, flat, quads, gd, hs, ho, idx, shard):
    prep = net.prepare_both(quads[idx], hs.take(idx), ho.take(idx), gd, shard=shard)
    loss = net.loss_prepared_both(prep)
    loss.backward()
    return float(loss)


def _scaling_worker(rank, world, port, out):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import cpu_abi_emulation
    cpu_abi_emulation.install()
    import ops
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    net, quads, gd, hs, ho, perm = _renet_setup()
    flat = parallel.FlatGrads(net)
    early = [p for n, p in net.named_parameters() if n in ('linear.weight', 'linear.bias')]
    red = parallel.OverlapReducer(flat, flat.span(('linear.weight', 'linear.bias'), net), early)
    ops.register_grad_done_hook(early, red.on_grad_done)
    res = {}
    for scaling in parallel.ScalingMode.MODES:
        mode = parallel.ScalingMode(scaling, 96, world)
        flat.zero()
        red.begin_step(head_passes=1, average=mode.average)
        loss = _mode_step(net, flat, quads, gd, hs, ho, mode.indices(perm, 1, rank), mode.shard(rank))
        red.finish()
        lsum = torch.tensor([loss])
        dist.all_reduce(lsum)
        res[scaling] = {'flat': flat.flat.clone(), 'loss_sum': float(lsum), 'global_batch': mode.global_batch,
                        'rank_batch': mode.rank_batch}
    torch.save(res, out % rank)
    dist.barrier()
    dist.destroy_process_group()


def test_three_scaling_modes_on_two_ranks_equal_their_single_process_definitions(tmp_path):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    world, port, out = 2, _free_port(), str(tmp_path / 'm%d.pt')
    mp.spawn(_scaling_worker, args=(world, port, out), nprocs=world, join=True)
    got = [torch.load(out % r) for r in range(world)]
    import cpu_abi_emulation
    undo = cpu_abi_emulation.install()
    try:
        net, quads, gd, hs, ho, perm = _renet_setup()
        flat = parallel.FlatGrads(net)
        for scaling in parallel.ScalingMode.MODES:
            assert torch.equal(got[0][scaling]['flat'], got[1][scaling]['flat']), scaling     # every rank ends with the same gradient
            mode = parallel.ScalingMode(scaling, 96, world)
            assert got[0][scaling]['global_batch'] == {'weak': 192, 'strong': 96, 'exact': 96}[scaling]
            assert got[0][scaling]['rank_batch'] == {'weak': 96, 'strong': 48, 'exact': 96}[scaling]
            if mode.exact:                     # == the ONE-rank step on the same reference batch: loss and gradient
                flat.zero()
                loss = _mode_step(net, flat, quads, gd, hs, ho, parallel.ScalingMode('exact', 96, 1).indices(perm, 1, 0), None)
                ref, ref_loss = flat.flat.clone(), loss
                assert abs(got[0][scaling]['loss_sum'] - ref_loss) <= 2e-6 * abs(ref_loss)
            else:                              # == the mean of the ranks' own batches
                acc, ref_loss = torch.zeros_like(flat.flat), 0.0
                for rank in range(world):
                    flat.zero()
                    ref_loss += _mode_step(net, flat, quads, gd, hs, ho, mode.indices(perm, 1, rank), None)
                    acc += flat.flat
                ref = acc / world
                assert abs(got[0][scaling]['loss_sum'] - ref_loss) <= 2e-6 * abs(ref_loss)
            scale = float(ref.abs().max())
            assert float((got[0][scaling]['flat'] - ref).abs().max()) <= 1e-5 * scale, scaling
    finally:
        undo()
