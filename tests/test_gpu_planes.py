"""GPU tests of the PLANES path (round 6): renet_pack_planes / renet_gemm_planes / renet_softmax_ce_planes and the score
head that runs on them -- against fp64 products, against the in-loop-split bf16x6 GEMM (same arithmetic: the results may
differ by fp32 summation order only), and against the fp32 softmax-CE kernel.  The planes GEMM replaces nn.Linear's forward
and backward GEMMs of the entity score head (reference model.py:89-91 and its autograd)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a HIP device'
    import renet_hip
    renet_hip.lib()
    return torch.device('cuda:0')


def _to(x, dev):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def test_pack_planes_is_the_three_term_rne_split(dev):
    """p1 = rne_bf16(x), p2 = rne_bf16(x - p1), p3 = rne_bf16(x - p1 - p2); padding zero; the ones column."""
    import renet_hip as K
    rng = np.random.RandomState(3)
    x = (rng.standard_normal((300, 77)) * np.exp(rng.uniform(-20, 20, (300, 77)))).astype(np.float32)
    xd = _to(x, dev)
    for ones in (False, True):
        m = K.pack_planes(xd, ones_col=ones)
        assert m.p.shape == (3, 512, 256) and (m.R, m.C) == (300, 77 + int(ones))
        p = K.planes_to_dense(m).cpu().numpy()
        r = torch.from_numpy(x)
        for k in range(3):
            t = r.to(torch.bfloat16).float()
            assert np.array_equal(p[k, :300, :77], t.numpy()), k
            r = r - t
        # three terms reproduce x to 2^-24 relative (usually exactly)
        rec = p[0, :300, :77].astype(np.float64) + p[1, :300, :77] + p[2, :300, :77]
        assert np.all(np.abs(rec - x) <= np.abs(x) * 2.0 ** -23)
        if ones:
            assert np.all(p[0, :300, 77] == 1.0) and np.all(p[1:, :300, 77] == 0.0)
            assert np.all(p[:, :300, 78:] == 0.0)
        else:
            assert np.all(p[:, :300, 77:] == 0.0)
        assert np.all(p[:, 300:, :] == 0.0)


CASES = [  # (m, n, k, ta, tb, split_k)
    (256, 128, 32, 0, 1, 1), (1, 1, 1, 0, 1, 1), (7, 5, 3, 0, 0, 1), (130, 257, 31, 0, 1, 1), (130, 257, 33, 0, 0, 1),
    (300, 129, 64, 1, 0, 1), (257, 130, 71, 1, 1, 1), (1024, 777, 600, 0, 1, 1), (1024, 777, 600, 0, 0, 1),
    (777, 600, 1024, 1, 0, 1), (96, 100, 5000, 1, 0, 7), (512, 600, 9000, 0, 0, 5), (2500, 2300, 96, 0, 1, 1),
    (200, 200, 4097, 1, 0, 3), (2048, 300, 200, 1, 1, 2),
]


@pytest.mark.parametrize('m,n,k,ta,tb,sk', CASES)
def test_gemm_planes_matches_fp64_and_the_in_loop_split(dev, m, n, k, ta, tb, sk):
    import renet_hip as K
    rng = np.random.RandomState(m * 131 + n * 17 + k + ta * 2 + tb)
    a = rng.uniform(-1, 1, (k, m) if ta else (m, k)).astype(np.float32)
    b = rng.uniform(-1, 1, (n, k) if tb else (k, n)).astype(np.float32)
    bias = rng.uniform(-1, 1, n).astype(np.float32)
    ref = (a.T if ta else a).astype(np.float64) @ (b.T if tb else b).astype(np.float64) + bias
    ad, bd = _to(a, dev), _to(b, dev)
    out = K.gemm_planes(K.pack_planes(ad), K.pack_planes(bd), ta=bool(ta), tb=bool(tb), bias=_to(bias, dev), split_k=sk)
    tol = 1e-5 * max(1, k) ** 0.5
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-5, atol=tol)
    # the in-loop split computes the same six products per element pair: equal up to fp32 summation order
    old = K.gemm(ad, bd, ta=bool(ta), tb=bool(tb), bias=_to(bias, dev), mode='bf16x6')
    np.testing.assert_allclose(out.cpu().numpy(), old.cpu().numpy(), rtol=2e-6, atol=2e-6 * max(1, k) ** 0.5)


_TILE_CHECK = r'''
import sys
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
import renet_hip as K
dev = torch.device('cuda:0')
for m, n, k, ta, tb, sk in %r:
    rng = np.random.RandomState(m + n + k)
    a = rng.uniform(-1, 1, (k, m) if ta else (m, k)).astype(np.float32)
    b = rng.uniform(-1, 1, (n, k) if tb else (k, n)).astype(np.float32)
    ref = (a.T if ta else a).astype(np.float64) @ (b.T if tb else b).astype(np.float64)
    out = K.gemm_planes(K.pack_planes(torch.from_numpy(a).to(dev)), K.pack_planes(torch.from_numpy(b).to(dev)),
                        ta=bool(ta), tb=bool(tb), split_k=sk)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-5, atol=1e-5 * max(1, k) ** 0.5)
print('ok')
'''


def test_gemm_planes_256_row_tile_variant(dev):
    """RENET_P6_TILE=256 selects the 8-wave 256 x 128 tile (one workgroup per CU) instead of the default 128 x 128 one:
    the same cases in a child process (the switch is read once)."""
    import os
    import subprocess
    import sys
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 're-net_amd')
    r = subprocess.run([sys.executable, '-c', _TILE_CHECK % (CASES,), pkg], env=dict(os.environ, RENET_P6_TILE='256'),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith('ok'), r.stdout + r.stderr


@pytest.mark.parametrize('m,n,k,sk', [(700, 130, 520, 1), (2100, 600, 300, 1), (600, 200, 5000, 6)])
def test_gemm_planes_epilogue_accumulate_device_alpha_and_bias_column(dev, m, n, k, sk):
    """C = beta * C + alpha * g * A^T [B | 1]: the ones column's product (column sums of A^T = the bias gradient) goes to
    col_out, the upstream scalar g is read from device memory -- the weight-gradient GEMM of the score head."""
    import renet_hip as K
    rng = np.random.RandomState(m + n + k)
    a = rng.uniform(-1, 1, (k, m)).astype(np.float32)            # stored [K, M]: ta
    b = rng.uniform(-1, 1, (k, n)).astype(np.float32)            # stored [K, N]
    c0 = rng.uniform(-1, 1, (m, n)).astype(np.float32)
    v0 = rng.uniform(-1, 1, m).astype(np.float32)
    g = np.float32(0.37)
    ref_c = c0 + 0.5 * g * (a.T.astype(np.float64) @ b.astype(np.float64))
    ref_v = v0 + 0.5 * g * a.T.astype(np.float64).sum(axis=1)
    c, v = _to(c0, dev), _to(v0, dev)
    K.gemm_planes(K.pack_planes(_to(a, dev)), K.pack_planes(_to(b, dev), ones_col=True), ta=True, out=c, col_out=v,
                  alpha=0.5, alpha_dev=torch.full((), float(g), device=dev), beta=1.0, split_k=sk)
    tol = 1e-5 * k ** 0.5
    np.testing.assert_allclose(c.cpu().numpy(), ref_c, rtol=1e-5, atol=tol)
    np.testing.assert_allclose(v.cpu().numpy(), ref_v, rtol=1e-5, atol=tol)


def test_gemm_planes_ignores_what_lies_outside_the_logical_matrices(dev):
    """Garbage in an output's row-stride padding and a strided output view are respected (ldc > N)."""
    import renet_hip as K
    rng = np.random.RandomState(9)
    a = rng.uniform(-1, 1, (515, 90)).astype(np.float32)
    b = rng.uniform(-1, 1, (333, 90)).astype(np.float32)
    buf = torch.full((515, 340), float('nan'), device=dev)
    out = buf[:, :333]
    K.gemm_planes(K.pack_planes(_to(a, dev)), K.pack_planes(_to(b, dev)), tb=True, out=out)
    np.testing.assert_allclose(out.cpu().numpy(), a.astype(np.float64) @ b.T.astype(np.float64), rtol=1e-5, atol=1e-4)
    assert torch.isnan(buf[:, 333:]).all()


@pytest.mark.parametrize('b,c', [(64, 23033), (33, 4100), (17, 2048), (9, 24576), (5, 1000), (3, 30000), (22, 12000),
                                 (130, 9000)])
def test_softmax_ce_planes_equals_the_fp32_kernel(dev, b, c):
    """Same losses as renet_softmax_ce; the planes sum to its gradient (each term the RNE of the running residual);
    padding columns zero.  Widths on both sides of the register kernel's range and unaligned rows."""
    import renet_hip as K
    rng = np.random.RandomState(b * 7 + c)
    x = (rng.standard_normal((b, c)) * 3).astype(np.float32)
    t = rng.randint(0, c, b).astype(np.int32)
    t[0], t[-1] = 0, c - 1
    ld = (c + 3) & ~3
    buf = torch.zeros(b, ld, device=dev)
    buf[:, :c] = _to(x, dev)
    logits = buf[:, :c]
    gs = 2.0 / b
    loss_p, dl = K.softmax_ce_planes(logits, _to(t, dev), gs)
    ref_in = _to(x, dev).clone()
    loss_f = K.softmax_ce(ref_in, _to(t, dev), gs, True)
    np.testing.assert_allclose(loss_p.cpu().numpy(), loss_f.cpu().numpy(), rtol=1e-6, atol=1e-6)
    p = K.planes_to_dense(dl).cpu().numpy()
    grad = ref_in.cpu().numpy()
    rec = p[0, :b, :c].astype(np.float64) + p[1, :b, :c] + p[2, :b, :c]
    np.testing.assert_allclose(rec, grad, rtol=2e-6, atol=1e-9)
    r = torch.from_numpy(rec.astype(np.float32))
    assert np.array_equal(p[0, :b, :c], r.to(torch.bfloat16).float().numpy()) or \
        np.mean(p[0, :b, :c] != r.to(torch.bfloat16).float().numpy()) < 1e-3      # (rec is a rounded sum: rare ties)
    assert np.all(p[:, :b, c:] == 0.0)                                 # the whole column padding of the written rows
    assert np.all(p[:, b:((b + 15) & ~15), :] == 0.0)                  # the k padding of dW (16 rows per half-stage)
    assert np.array_equal(buf[:, :c].cpu().numpy(), x)               # the logits are left untouched


def _head_grads(dev, planes, b=300, d=100, n_ent=5000, seed=4, drop=0.0):
    import ops
    import renet_hip as K
    old = K.PLANES
    K.PLANES = planes
    try:
        rng = np.random.RandomState(seed)
        ent = torch.nn.Parameter(_to(rng.uniform(-.5, .5, (n_ent, d)).astype(np.float32), dev))
        rel = torch.nn.Parameter(_to(rng.uniform(-.5, .5, (40, d)).astype(np.float32), dev))
        w = torch.nn.Parameter(_to(rng.uniform(-.1, .1, (n_ent, 3 * d)).astype(np.float32), dev))
        bias = torch.nn.Parameter(_to(rng.uniform(-.1, .1, n_ent).astype(np.float32), dev))
        h = _to(rng.uniform(-1, 1, (b, d)).astype(np.float32), dev).requires_grad_(True)
        ia = _to(rng.randint(0, n_ent, b).astype(np.int32), dev)
        ic = _to(rng.randint(0, 40, b).astype(np.int32), dev)
        tgt = _to(rng.randint(0, n_ent, b).astype(np.int32), dev)
        import graph as G

        def plan(idx):
            p = G.SegPlan.host(idx.cpu().numpy().astype(np.int64))
            for f in ('order', 'seg_ptr', 'target'):
                setattr(p, f, torch.from_numpy(getattr(p, f)).to(dev))
            return p
        loss = ops.HeadCEFn.apply(ent, ia, h, rel, ic, w, bias, tgt, plan(ia), plan(ic), drop, 12345)
        (loss * 0.7).backward()
        torch.cuda.synchronize()
        return [loss.item()] + [t.grad.cpu().numpy() for t in (ent, rel, w, bias, h)]
    finally:
        K.PLANES = old


def test_score_head_on_planes_equals_the_in_loop_split_head(dev):
    """ops.HeadCEFn with the planes GEMMs (wide class dimension) against the same Function on the in-loop split: loss and
    every gradient (entity / relation embeddings through the scatter-adds, weight, bias -- which comes out of the dW GEMM's
    ones column -- and the hidden state), upstream scalar 0.7 applied through alpha_dev."""
    a = _head_grads(dev, True)
    b = _head_grads(dev, False)
    assert abs(a[0] - b[0]) < 1e-6 * abs(b[0])
    for x, y, name in zip(a[1:], b[1:], ('ent', 'rel', 'weight', 'bias', 'h')):
        np.testing.assert_allclose(x, y, rtol=1e-4, atol=1e-6 * max(1.0, float(np.abs(y).max()) * 10), err_msg=name)


_SOFTMAX_ROWS_CHECK = '''
import sys
sys.path.insert(0, sys.argv[1])
sys.path.insert(0, sys.argv[2])
import torch
import test_gpu_planes as T
dev = torch.device('cuda:0')
for b, c in ((64, 23033), (33, 4100), (9, 24576)):
    T.test_softmax_ce_planes_equals_the_fp32_kernel.__wrapped__(dev, b, c) if hasattr(
        T.test_softmax_ce_planes_equals_the_fp32_kernel, '__wrapped__') else T.test_softmax_ce_planes_equals_the_fp32_kernel(dev, b, c)
print('ok')
'''


def test_softmax_ce_planes_one_row_per_workgroup_variant(dev):
    """RENET_SOFTMAX_ROWS=1 selects the first planes writer (one row per workgroup, 32-byte pieces) instead of the default
    four-row one (whole lines through LDS): the same checks in a child process (the switch is read once)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    pkg = os.path.join(os.path.dirname(here), 're-net_amd')
    r = subprocess.run([sys.executable, '-c', _SOFTMAX_ROWS_CHECK, pkg, here], env=dict(os.environ, RENET_SOFTMAX_ROWS='1'),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith('ok'), r.stdout + r.stderr
