"""GPU tests of the bf16 mixed-precision mode (BASELINE config 5: YAGO, n_hidden=400 bf16, seq_len=15):
renet_gemm_bf16 rounds every fp32 GEMM operand to bf16 (RNE) on its way into LDS, multiplies on
v_mfma_f32_32x32x16_bf16 and accumulates / stores in fp32; every other kernel of the step is unchanged.

Tolerances against the fp32 reference outputs (the reference has no bf16 path: they are OURS, stated here and in
DESIGN.md): a bf16 operand carries 8 significant bits (relative rounding error <= 2^-9), so
  * a GEMM of K-long dot products:           |C - C_fp64| <= 2^-8 * (3 sqrt(K) + 1) * max|a| * max|b|  (random walk
                                             of 2K roundings of <= 2^-9 each, 3 sigma; >= the worst case for K <= 9)
  * training step at config 5:               losses 2e-3 relative; h_n / logits 2e-2 of the tensor's max |value|;
                                             gradients 0.15 of the tensor's max |value| per entry (the deepest
                                             ones, rgcn1.*, pass through ~10 bf16 GEMMs) and 5e-2 in Frobenius norm.
"""
import numpy as np
import pytest
import torch

from helpers import load_golden
from oracle import config_cases as C

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a HIP device'
    import renet_hip
    renet_hip.lib()
    return torch.device('cuda:0')


@pytest.mark.parametrize('m,n,k', [(1, 1, 1), (7, 5, 3), (128, 128, 32), (257, 130, 71), (1024, 777, 600), (333, 400, 4),
                                   (96, 100, 5000)])
@pytest.mark.parametrize('ta,tb', [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_bf16_gemm_matches_fp64_of_the_rounded_operands(dev, m, n, k, ta, tb):
    import renet_hip as K
    rng = np.random.RandomState(m * 131 + n * 17 + k + ta * 2 + tb)
    a = rng.uniform(-1, 1, (k, m) if ta else (m, k)).astype(np.float32)
    b = rng.uniform(-1, 1, (n, k) if tb else (k, n)).astype(np.float32)
    bias = rng.uniform(-1, 1, n).astype(np.float32)
    ta_, tb_ = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    sk = 7 if k >= 5000 else None
    out = K.gemm(ta_, tb_, ta=bool(ta), tb=bool(tb), bias=torch.from_numpy(bias).to(dev), mode='bf16', split_k=sk)
    # exact statement: the kernel multiplies the bf16-rounded operands exactly and sums in fp32
    ar = ta_.bfloat16().double().cpu().numpy()
    br = tb_.bfloat16().double().cpu().numpy()
    ref = (ar.T if ta else ar) @ (br.T if tb else br) + bias
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-5, atol=2e-6 * max(1, k))
    # and the distance to the unrounded product stays inside the stated bf16 bound
    full = (a.T if ta else a).astype(np.float64) @ (b.T if tb else b).astype(np.float64) + bias
    assert np.abs(out.cpu().numpy() - full).max() <= 2.0 ** -8 * (3 * np.sqrt(k) + 1) + 1e-6


def test_config5_training_step_in_bf16_mode(dev):
    """ONE eval-mode training step at config 5's sizes (YAGO-shaped, n_hidden 400, seq_len 15, B 1024) with the
    GEMMs in bf16 mode, against the UNMODIFIED reference's fp32 outputs (tests/golden/config_yago_d400_l15.npz)."""
    import renet_hip as K
    import preprocess as P
    from test_gpu_config import hip_step
    gold = load_golden('config_yago_d400_l15.npz')
    case = C.build_case('yago_d400_l15', gold=gold)
    quads, idx, L = case['quads'], case['idx'], case['spec']['seq_len']
    fh = {t: P.HistoryIndex(quads, t, history_len=L).take(idx) for t in ('s', 'o')}
    taps = []
    old = K.GEMM_MODE
    K.GEMM_MODE = 'bf16'
    try:
        net, loss_s, loss_o = hip_step(case, dev, fh['s'], fh['o'], tap=lambda n, t: taps.append((n, t.detach().clone())))
    finally:
        K.GEMM_MODE = old
    report = {}
    for tag, loss in (('s', loss_s), ('o', loss_o)):
        ref = float(gold['loss_' + tag])
        report['loss_' + tag] = abs(loss.item() - ref) / abs(ref)
        assert report['loss_' + tag] < 2e-3, (tag, loss.item(), ref)
    per_dir = {'s': taps[:4], 'o': taps[4:8]}
    for tag in ('s', 'o'):
        lens = np.diff(np.asarray(gold['hist_%s_seq_ptr' % tag]))
        perm = np.argsort(-lens, kind='stable')
        for key, t in (('h_n', per_dir[tag][0][1]), ('logits', per_dir[tag][2][1])):
            a = t.cpu().numpy()
            full = np.zeros_like(a)
            full[perm] = a
            ok, err, scale = C.compare_packed(gold, '%s_%s' % (tag, key), full, rel=2e-2)
            report['%s_%s' % (tag, key)] = err / scale
            assert ok, (tag, key, err, scale)
    worst = 0.0
    for k, p in net.named_parameters():
        g = p.grad.cpu().numpy()
        ref_s = np.asarray(gold['grad.' + k + '__samp']) if ('grad.' + k + '__samp') in gold else np.asarray(gold['grad.' + k]).reshape(-1)
        from oracle import fixtures
        got_s = g.reshape(-1)[fixtures.sample_idx(g.size)] if ('grad.' + k + '__samp') in gold else g.reshape(-1)
        scale = float(np.abs(ref_s).max())
        err = float(np.abs(got_s - ref_s).max())
        worst = max(worst, err / scale)
        assert err <= 0.15 * scale, (k, err, scale)
        if ('grad.' + k + '__norm') in gold:
            nr = float(gold['grad.' + k + '__norm'])
            assert abs(float(np.linalg.norm(g.astype(np.float64))) - nr) <= 5e-2 * nr, k
    report['worst_grad'] = worst
    print('bf16 config-5 deviations from the fp32 reference:', {k: float('%.3g' % v) for k, v in report.items()})
