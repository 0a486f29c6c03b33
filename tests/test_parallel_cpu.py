"""World-size-2 `gloo` test (CPU) of the data-parallel exchange: flat gradient views + one all-reduce
== single-process gradient accumulation over the same two batches (re-net_amd/parallel.py)."""
import os
import socket

import numpy as np
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
