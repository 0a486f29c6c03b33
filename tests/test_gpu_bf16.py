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
                                   (96, 100, 5000),
                                   (6000, 2000, 130)])      # >= 256 tiles of 256 x 128: the tall bf16s kernel
@pytest.mark.parametrize('ta,tb', [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize('mode', ['bf16', 'bf16s'])
def test_bf16_gemm_matches_fp64_of_the_rounded_operands(dev, m, n, k, ta, tb, mode):
    import renet_hip as K
    rng = np.random.RandomState(m * 131 + n * 17 + k + ta * 2 + tb)
    a = rng.uniform(-1, 1, (k, m) if ta else (m, k)).astype(np.float32)
    b = rng.uniform(-1, 1, (n, k) if tb else (k, n)).astype(np.float32)
    bias = rng.uniform(-1, 1, n).astype(np.float32)
    ta_, tb_ = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
    sk = 7 if k >= 5000 else None
    # 'bf16': fp32 tensors, rounded inside the kernel; 'bf16s': bf16 STORAGE (packed operands, LDS-DMA kernel)
    out = K.gemm(ta_, tb_, ta=bool(ta), tb=bool(tb), bias=torch.from_numpy(bias).to(dev), mode=mode, split_k=sk)
    # exact statement: the kernel multiplies the bf16-rounded operands exactly and sums in fp32
    ar = ta_.bfloat16().double().cpu().numpy()
    br = tb_.bfloat16().double().cpu().numpy()
    ref = (ar.T if ta else ar) @ (br.T if tb else br) + bias
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-5, atol=2e-6 * max(1, k))
    # and the distance to the unrounded product stays inside the stated bf16 bound
    full = (a.T if ta else a).astype(np.float64) @ (b.T if tb else b).astype(np.float64) + bias
    assert np.abs(out.cpu().numpy() - full).max() <= 2.0 ** -8 * (3 * np.sqrt(k) + 1) + 1e-6


def test_bf16_storage_gemm_on_prepacked_operands_and_sliced_weights(dev):
    """renet_gemm_bf16s on operands packed ONCE and consumed in both roles (K-contiguous and K-strided), with beta
    accumulation into a strided output view, and a leading column block of a packed weight (W_ih[:, :live])."""
    import renet_hip as K
    rng = np.random.RandomState(3)
    m, n, k = 700, 1200, 1600
    a = torch.from_numpy(rng.uniform(-1, 1, (m, k)).astype(np.float32)).to(dev)
    w = torch.from_numpy(rng.uniform(-1, 1, (n, k)).astype(np.float32)).to(dev)       # nn.Linear layout [N, K]
    pa, pw = K.pack_bf16(a), K.pack_bf16(w)
    ar, wr = a.bfloat16().double(), w.bfloat16().double()
    y = K.gemm(pa, pw, tb=True)                                                       # a @ w^T
    torch.testing.assert_close(y.double(), ar @ wr.t(), rtol=1e-5, atol=2e-6 * k)
    gy = torch.from_numpy(rng.uniform(-1, 1, (m, n)).astype(np.float32)).to(dev)
    pg = K.pack_bf16(gy)
    gr = gy.bfloat16().double()
    da = K.gemm(pg, pw)                                                               # gy @ w (w K-strided)
    torch.testing.assert_close(da.double(), gr @ wr, rtol=1e-5, atol=2e-6 * n)
    dw0 = torch.from_numpy(rng.uniform(-1, 1, (n, k + 8)).astype(np.float32)).to(dev)
    dw = dw0.clone()
    K.gemm(pg, pa, ta=True, out=dw[:, 4:4 + k], beta=1.0)                             # gy^T @ a, both K-strided
    torch.testing.assert_close(dw[:, 4:4 + k].double(), gr.t() @ ar + dw0[:, 4:4 + k].double(), rtol=1e-5, atol=2e-6 * m)
    assert torch.equal(dw[:, :4], dw0[:, :4]) and torch.equal(dw[:, 4 + k:], dw0[:, 4 + k:])
    # registered weight: cached copy, and a leading column block of it (contraction over 1200 of its 1600 columns
    # would need a 64-aligned end: 1216 is; 1200 is not -> packed afresh, same result)
    K.register_weights([w])
    try:
        for live in (1216, 1200, 640):
            d = K.gemm(gy, w[:, :live], mode='bf16s')                                 # [m, live] = gy @ w[:, :live]
            torch.testing.assert_close(d.double(), gr @ wr[:, :live], rtol=1e-5, atol=2e-6 * n)
            x = torch.from_numpy(rng.uniform(-1, 1, (m, live)).astype(np.float32)).to(dev)
            z = K.gemm(x, w[:, :live], tb=True, mode='bf16s')                         # contraction over the block
            torch.testing.assert_close(z.double(), x.bfloat16().double() @ wr[:, :live].t(), rtol=1e-5, atol=2e-6 * live)
        w.mul_(2.0)
        K.weights_changed()
        d = K.gemm(gy, w, mode='bf16s')
        torch.testing.assert_close(d.double(), gr @ (2 * wr), rtol=1e-5, atol=4e-6 * n)
    finally:
        K.unregister_weights([w])


# per-tensor relative L2 bound of the bf16-mode gradients against the fp32 reference: 2 x the largest value observed on MI355X
# (profiles/r05_*_gpu_tests.txt prints every tensor's value); REL_L2_BOUND overrides it per tensor where needed
REL_L2_DEFAULT = 6.6e-3            # observed (profiles/r05_g_gpu_tests.txt, both bf16 modes): <= 3.3e-3 for 17 of the 20 tensors
REL_L2_BOUND = {'ent_embeds': 4e-2,                       # 1.95e-2: the entity rows are READ as bf16 by layer 1 (bf16 storage)
                'aggregator.rgcn1.weight': 5.6e-2,        # 2.78e-2  } layer 1 multiplies bf16 entity rows by bf16 relation
                'aggregator.rgcn1.loop_weight': 6.4e-2}   # 3.19e-2  } blocks: its weight gradients carry both roundings


@pytest.mark.parametrize('mode', ['bf16', 'bf16s'])
def test_config5_training_step_in_bf16_mode(dev, mode):
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
    K.GEMM_MODE = mode
    try:
        net, loss_s, loss_o = hip_step(case, dev, fh['s'], fh['o'], tap=lambda n, t: taps.append((n, t.detach().clone())))
    finally:
        K.GEMM_MODE = old
    report = {}
    for tag, loss in (('s', loss_s), ('o', loss_o)):
        ref = float(gold['loss_' + tag])
        report['loss_' + tag] = abs(loss.item() - ref) / abs(ref)
        assert report['loss_' + tag] < 5e-5, (tag, loss.item(), ref)          # observed 4e-7 / 6e-7 (bench: 4e-6 - 1.3e-5)
    per_dir = {'s': taps[:4], 'o': taps[4:8]}
    for tag in ('s', 'o'):
        lens = np.diff(np.asarray(gold['hist_%s_seq_ptr' % tag]))
        perm = np.argsort(-lens, kind='stable')
        for key, t in (('h_n', per_dir[tag][0][1]), ('logits', per_dir[tag][2][1])):
            a = t.cpu().numpy()
            full = np.zeros_like(a)
            full[perm] = a
            ok, err, scale = C.compare_packed(gold, '%s_%s' % (tag, key), full, rel=8e-3)      # observed 2.7e-3 of max
            report['%s_%s' % (tag, key)] = err / scale
            assert ok, (tag, key, err, scale)
    worst = 0.0
    rel_l2 = {}
    for k, p in net.named_parameters():
        g = p.grad.cpu().numpy()
        ref_s = np.asarray(gold['grad.' + k + '__samp']) if ('grad.' + k + '__samp') in gold else np.asarray(gold['grad.' + k]).reshape(-1)
        from oracle import fixtures
        got_s = g.reshape(-1)[fixtures.sample_idx(g.size)] if ('grad.' + k + '__samp') in gold else g.reshape(-1)
        scale = float(np.abs(ref_s).max())
        err = float(np.abs(got_s - ref_s).max())
        worst = max(worst, err / scale)
        assert err <= 0.15 * scale, (k, err, scale)                 # observed worst entry 7.3 % of its tensor's max: 2x
        # the sharper statement (review r4): per-tensor RELATIVE L2 error over the compared positions
        rel_l2[k] = float(np.linalg.norm((got_s - ref_s).astype(np.float64)) / max(np.linalg.norm(ref_s.astype(np.float64)), 1e-30))
        assert rel_l2[k] <= REL_L2_BOUND.get(k, REL_L2_DEFAULT), (k, rel_l2[k])
        if ('grad.' + k + '__norm') in gold:
            nr = float(gold['grad.' + k + '__norm'])
            assert abs(float(np.linalg.norm(g.astype(np.float64))) - nr) <= 5e-2 * nr, k
    report['worst_grad'] = worst
    print('bf16 config-5 per-tensor relative L2 gradient error [%s]:' % mode, {k: float('%.3g' % v) for k, v in rel_l2.items()})
    print('bf16 config-5 deviations from the fp32 reference:', {k: float('%.3g' % v) for k, v in report.items()})


@pytest.mark.parametrize('i,h,nseq', [(1600, 400, 45), (800, 200, 31), (300, 100, 20)])
def test_gru_in_bf16_mode_matches_torch_gru_within_the_bf16_bound(dev, i, h, nseq):
    """bf16 mode of the recurrence (renet_gru_{fwd,bwd}_layouts_bf16: W_hh and the state rounded to bf16 as MFMA
    operands, one product per pair; fp32 accumulation, gate math and state) + bf16-storage input projection against
    torch.nn.GRU in fp32.  Bounds (ours; the reference has no bf16 path): h_n 2e-2 of its max, dX and every parameter
    gradient 6e-2 of the tensor's max."""
    import model as M
    import renet_hip as K
    torch.manual_seed(11)
    lens = [10] * (nseq - 11) + [9, 9, 7, 5, 5, 5, 3, 2, 1, 1, 1]
    b, l = len(lens), 10
    ref = torch.nn.GRU(i, h, batch_first=True)
    x = torch.randn(b, l, i) * 0.5
    for k, n in enumerate(lens):
        x[k, n:] = 0
    x.requires_grad_(True)
    packed = torch.nn.utils.rnn.pack_padded_sequence(x, lens, batch_first=True)
    _, hn = ref(packed)
    gout = torch.randn(b, h)
    (hn[0] * gout).sum().backward()
    mine = M.GRU(i, h).to(dev)
    mine.load_state_dict(ref.state_dict())
    xd = packed.data.detach().to(dev).requires_grad_(True)
    pk = torch.nn.utils.rnn.PackedSequence(xd, packed.batch_sizes)
    old = K.GEMM_MODE
    K.GEMM_MODE = 'bf16s'
    try:
        _, hm = mine(pk, total_rows=b)
        (hm[0, :b] * gout.to(dev)).sum().backward()
        torch.cuda.synchronize()
    finally:
        K.GEMM_MODE = old

    def close(a, r, frac, what):
        a, r = a.detach().cpu().double().numpy(), r.detach().double().numpy()
        err, scale = np.abs(a - r).max(), np.abs(r).max()
        assert err <= frac * scale, (what, err, scale)
    close(hm[0, :b], hn[0], 2e-2, 'h_n')
    dx_ref = torch.nn.utils.rnn.pack_padded_sequence(x.grad, lens, batch_first=True).data
    close(xd.grad, dx_ref, 6e-2, 'dX')
    for name, p in mine.named_parameters():
        close(p.grad, getattr(ref, name).grad, 6e-2, name)
