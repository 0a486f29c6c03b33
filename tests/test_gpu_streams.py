"""ops.DualHeadCEFn and the fork/join side stream (ops._Side): same values as the single-stream, two-Function path."""
import os
import sys

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


def _device_plan(idx, dev):
    import graph as G
    h = G.SegPlan.host(idx)
    p = G.SegPlan()
    p.order, p.seg_ptr, p.target = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in
                                    (h.order, h.seg_ptr, h.target))
    p.num_segments = h.num_segments
    return p


def _head_case(dev, seed, b=384, n_ent=1500, n_rel=46, d=200):
    rng = np.random.RandomState(seed)
    f32 = lambda a: torch.from_numpy(np.asarray(a, np.float32)).to(dev)       # noqa: E731
    i32 = lambda a: torch.from_numpy(np.asarray(a, np.int32)).to(dev)         # noqa: E731
    ia = rng.zipf(1.3, b) % n_ent                                              # hot subjects: long segments
    ic = rng.randint(0, n_rel, b)
    case = dict(
        ent=f32(rng.randn(n_ent, d) * 0.3), rel=f32(rng.randn(n_rel, d) * 0.3),
        h1=f32(rng.randn(b, d) * 0.5), h2=f32(rng.randn(b, d) * 0.5),
        w1=f32(rng.randn(n_ent, 3 * d) * 0.05), b1=f32(rng.randn(n_ent) * 0.1),
        w2=f32(rng.randn(n_rel, 2 * d) * 0.05), b2=f32(rng.randn(n_rel) * 0.1),
        ia=i32(ia), ic=i32(ic), t1=i32(rng.randint(0, n_ent, b)), t2=i32(rng.randint(0, n_rel, b)),
        plan_a=_device_plan(ia, dev), plan_c=_device_plan(ic, dev))
    return case


def _run_heads(case, dual, drop_p, scale, with_grad_buffers):
    import ops
    names = ('ent', 'rel', 'h1', 'h2', 'w1', 'b1', 'w2', 'b2')
    leaves = {n: case[n].clone().requires_grad_(True) for n in names}
    if with_grad_buffers:                        # in-place accumulation targets, as under parallel.HipAdam
        for n in ('ent', 'rel', 'w1', 'b1', 'w2', 'b2'):
            leaves[n].grad = torch.full_like(leaves[n], 0.25)
    L = leaves
    if dual:
        loss = ops.DualHeadCEFn.apply(L['ent'], case['ia'], L['h1'], L['rel'], case['ic'], L['w1'], L['b1'], case['t1'],
                                      L['h2'], L['w2'], L['b2'], case['t2'], case['plan_a'], case['plan_c'], drop_p,
                                      11, 12, scale, 0.1)
    else:
        l1 = ops.HeadCEFn.apply(L['ent'], case['ia'], L['h1'], L['rel'], case['ic'], L['w1'], L['b1'], case['t1'],
                                case['plan_a'], case['plan_c'], drop_p, 11, scale)
        l2 = ops.HeadCEFn.apply(L['ent'], case['ia'], L['h2'], None, None, L['w2'], L['b2'], case['t2'],
                                case['plan_a'], None, drop_p, 12, scale)
        loss = l1 + 0.1 * l2
    loss.backward()
    torch.cuda.synchronize()
    return loss.item(), {n: L[n].grad.detach().cpu().double().numpy() for n in names}


@pytest.mark.parametrize('drop_p', [0.0, 0.5])
@pytest.mark.parametrize('with_grad_buffers', [False, True])
def test_dual_head_equals_two_single_heads(dev, drop_p, with_grad_buffers):
    case = _head_case(dev, 3)
    l_ref, g_ref = _run_heads(case, False, drop_p, 2.0, with_grad_buffers)
    l_new, g_new = _run_heads(case, True, drop_p, 2.0, with_grad_buffers)
    assert abs(l_new - l_ref) <= 2e-6 * abs(l_ref), (l_new, l_ref)
    for n in g_ref:
        scale = np.abs(g_ref[n]).max()
        assert scale > 0
        # same kernels on the same inputs; the relation head's 0.1 is folded into its CE gradient before the GEMMs
        # instead of scaling it afterwards, and the two heads' ent[s] row gradients are added before the scatter
        assert np.abs(g_new[n] - g_ref[n]).max() <= 3e-6 * scale, (n, np.abs(g_new[n] - g_ref[n]).max(), scale)


def test_side_stream_changes_no_value(dev, monkeypatch):
    """One merged training step with the fork/join streams and with RENET_SIDE_STREAM=0: bit-identical loss and flat
    gradient (every kernel sees the same inputs; only the launch interleaving differs)."""
    import model as M
    import ops
    import parallel
    import preprocess as P
    import synth
    quads, num_ent, num_rels, _ = synth.make_stream('ICEWS18', seed=5, num_t=40)
    gd = P.build_graph_dict(quads, num_rels)
    hs, ho = P.HistoryIndex(quads, 's', 10), P.HistoryIndex(quads, 'o', 10)
    idx = np.random.RandomState(1).permutation(len(quads))[:256]
    results = []
    for side in (True, False):
        monkeypatch.setattr(ops, 'SIDE_STREAM', side)
        torch.manual_seed(7)
        ops.reset_seed_counter()
        net = M.RENet(num_ent, 200, num_rels, dropout=0.5, seq_len=10, num_k=10)
        gen = torch.Generator().manual_seed(3)
        net.global_emb = {int(t): torch.randn(1, 1, 200, generator=gen) * 0.1 for t in gd}
        net.to(dev).train()
        opt = parallel.HipAdam(net, lr=1e-3, weight_decay=1e-5, max_norm=1.0)
        fs, fo = hs.take(idx), ho.take(idx)
        losses = []
        for _ in range(2):
            prep = net.prepare_both(quads[idx], fs, fo, gd)
            loss = net.loss_prepared_both(prep)
            loss.backward()
            losses.append(loss.item())
            flat = opt.grads.flat.detach().clone()
            opt.step()
        torch.cuda.synchronize()
        results.append((losses, flat, torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()))
        opt.close()
    (la, fa, pa), (lb, fb, pb) = results
    assert la == lb, (la, lb)
    assert float(fa.abs().max()) > 0
    assert torch.equal(fa, fb)
    assert torch.equal(pa, pb)


def test_two_models_with_different_gemm_modes_in_one_process(dev, monkeypatch):
    """Review r3 (weak 8): the fp32-class GEMM mode was one process-wide switch.  Since round 4 a model carries its own
    (`net.gemm_mode = 'f16x3' | 'bf16x6' | None`), forward and backward run inside that scope (autograd Functions remember
    the mode of their forward pass) and HipAdam measures the f16x3 weight bounds under it.  Two models with different
    modes, stepped ALTERNATELY in one process, must each end bit-identical to the same model trained alone with that mode
    as the process default."""
    import model as M
    import ops
    import parallel
    import preprocess as P
    import renet_hip as K
    import synth
    if K.GEMM_MODE not in ('bf16x6', 'f16x3'):
        pytest.skip('this test alternates the two split modes')
    quads, num_ent, num_rels, _ = synth.make_stream('ICEWS18', seed=5, num_t=40)
    gd = P.build_graph_dict(quads, num_rels)
    hs, ho = P.HistoryIndex(quads, 's', 10), P.HistoryIndex(quads, 'o', 10)
    perm = np.random.RandomState(1).permutation(len(quads))

    def make(mode):
        torch.manual_seed(7)
        net = M.RENet(num_ent, 200, num_rels, dropout=0.5, seq_len=10, num_k=10)
        gen = torch.Generator().manual_seed(3)
        net.global_emb = {int(t): torch.randn(1, 1, 200, generator=gen) * 0.1 for t in gd}
        net.to(dev).train()
        net.gemm_mode = mode
        return net, parallel.HipAdam(net, lr=1e-3, weight_decay=1e-5, max_norm=1.0)

    def step(net, opt, k, seed_base):
        idx = perm[k * 512:(k + 1) * 512]
        ops.reset_seed_counter(seed_base + 100 * k)            # same dropout masks whoever else runs in between
        with opt.step_scope(head_passes=1):
            loss = net.loss_prepared_both(net.prepare_both(quads[idx], hs.take(idx), ho.take(idx), gd))
            loss.backward()
            opt.step()
        return loss.item()

    def flat(net):
        return torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()

    # alternately, each with its own mode
    (na, oa), (nb, ob) = make('f16x3'), make('bf16x6')
    la, lb = [], []
    for k in range(3):
        la.append(step(na, oa, k, 1000))
        lb.append(step(nb, ob, k, 5000))
    torch.cuda.synchronize()
    pa, pb = flat(na), flat(nb)
    oa.close(); ob.close()
    # alone, the mode as the process default
    solo = {}
    for mode, base in (('f16x3', 1000), ('bf16x6', 5000)):
        monkeypatch.setattr(K, 'GEMM_MODE', mode)
        net, opt = make(None)
        solo[mode] = ([step(net, opt, k, base) for k in range(3)], flat(net))
        opt.close()
    assert la == solo['f16x3'][0] and lb == solo['bf16x6'][0], (la, solo['f16x3'][0], lb, solo['bf16x6'][0])
    assert torch.equal(pa, solo['f16x3'][1]) and torch.equal(pb, solo['bf16x6'][1])
    assert not torch.equal(pa, pb)                              # (the two modes do differ in the last bits)
    with pytest.raises(K.RenetHipError):
        with K.gemm_mode('bf16s'):                              # the storage mode stays process-wide
            pass


def test_exact_fp32_mode_per_model_equals_the_process_wide_mode(dev):
    """Round 5: `net.gemm_mode = 'f32'` (exact fp32 MFMA products in the GEMMs AND in the GRU recurrences, through the
    renet_gru_*_layouts_f32 entries) inside a bf16x6 process must train bit-identically to a process started with RENET_GEMM=f32
    -- until round 4 the exact mode existed as a process-wide switch only (the library read RENET_GEMM for the recurrences)."""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = {}
    for tag, args, env in (('per_model', ['f32'], {'RENET_GEMM': 'bf16x6'}), ('process', [], {'RENET_GEMM': 'f32'}),
                           ('split', [], {'RENET_GEMM': 'bf16x6'})):
        r = subprocess.run([sys.executable, os.path.join(here, 'mode_run.py')] + args, env=dict(os.environ, **env),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
        out[tag] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert out['per_model']['process_default'] == 'bf16x6' and out['process']['process_default'] == 'f32'
    assert out['per_model']['losses'] == out['process']['losses'], (out['per_model'], out['process'])
    assert out['per_model']['digest'] == out['process']['digest']
    assert out['split']['digest'] != out['process']['digest']            # (the split mode does differ in the last bits)


def test_deferred_weight_gradients_change_no_value(dev, monkeypatch):
    """Inside `with opt.step_scope(...)` the GRU parameter-gradient GEMMs run on the side stream and are joined only in
    opt.step() (ops.DEFER_WEIGHT_GRADS, round 4): three declared steps -- merged pass and the pair of passes -- against
    the same steps with RENET_DEFER_GRADS=0: bit-identical losses and parameters (same kernels, same order per gradient
    buffer; only the interleaving with the rest of the backward pass differs).  The intermediate tensors of the backward
    pass are freed while the side stream may still read them: record_stream must keep the allocator from recycling
    them early -- a wrong value here would show up as a mismatch."""
    import model as M
    import ops
    import parallel
    import preprocess as P
    import synth
    quads, num_ent, num_rels, _ = synth.make_stream('ICEWS18', seed=5, num_t=40)
    gd = P.build_graph_dict(quads, num_rels)
    hs, ho = P.HistoryIndex(quads, 's', 10), P.HistoryIndex(quads, 'o', 10)
    perm = np.random.RandomState(1).permutation(len(quads))
    for passes in ('merged', 'pair'):
        results = []
        for defer in ('1', '0'):
            monkeypatch.setenv('RENET_DEFER_GRADS', defer)
            torch.manual_seed(7)
            ops.reset_seed_counter()
            net = M.RENet(num_ent, 200, num_rels, dropout=0.5, seq_len=10, num_k=10)
            gen = torch.Generator().manual_seed(3)
            net.global_emb = {int(t): torch.randn(1, 1, 200, generator=gen) * 0.1 for t in gd}
            net.to(dev).train()
            opt = parallel.HipAdam(net, lr=1e-3, weight_decay=1e-5, max_norm=1.0)
            losses = []
            for k in range(3):
                idx = perm[k * 512:(k + 1) * 512]
                fs, fo = hs.take(idx), ho.take(idx)
                with opt.step_scope(head_passes=1 if passes == 'merged' else 2):
                    if passes == 'merged':
                        loss = net.loss_prepared_both(net.prepare_both(quads[idx], fs, fo, gd))
                    else:
                        loss = net.loss_prepared_pair(net.prepare(quads[idx], fs, gd, subject=True),
                                                      net.prepare(quads[idx], fo, gd, subject=False))
                    loss.backward()
                    # churn the allocator while the deferred kernels may still be running
                    junk = [torch.full((n_, 600), float('nan'), device=dev) for n_ in (4000, 9000, 16000)]
                    del junk
                    opt.step()
                losses.append(loss.item())
            torch.cuda.synchronize()
            assert not ops._deferred
            results.append((losses, torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()))
            opt.close()
        (la, pa), (lb, pb) = results
        assert la == lb, (passes, la, lb)
        assert torch.isfinite(pa).all()
        assert torch.equal(pa, pb), passes


@pytest.mark.parametrize('d', [100, 200, 400])
@pytest.mark.parametrize('n_rows,n_tgt,zipf', [(5000, 900, 1.2), (40, 4000, 0.0), (3000, 3, 0.0), (1, 1, 0.0),
                                               (70000, 23033, 1.1)])
def test_segment_add_matches_fp64_scatter_add(dev, d, n_rows, n_tgt, zipf):
    """renet_segment_add / renet_segment_add2: short segments (a wave each), long segments (all 16 waves of the
    workgroup), targets without rows, one-row inputs; against a float64 index_add and bit-reproducible."""
    import renet_hip as K
    rng = np.random.RandomState(n_rows + d)
    idx = (rng.zipf(zipf, n_rows) % n_tgt) if zipf else rng.randint(0, n_tgt, n_rows)
    plan = _device_plan(idx, dev)
    src0 = torch.from_numpy(rng.randn(n_rows, d).astype(np.float32)).to(dev)
    src1 = torch.from_numpy(rng.randn(n_rows, d).astype(np.float32)).to(dev)
    base = torch.from_numpy(rng.randn(n_tgt, d).astype(np.float32)).to(dev)
    lidx = torch.from_numpy(idx.astype(np.int64)).to(dev)
    want0 = base.double().index_add(0, lidx, src0.double())
    want1 = base.double().index_add(0, lidx, src1.double())
    got = K.segment_add(src0, plan, base.clone())
    a, b = base.clone(), base.clone()
    K.segment_add2(src0, src1, plan, a, b)
    again = K.segment_add(src0, plan, base.clone())
    torch.cuda.synchronize()
    longest = int(np.bincount(idx).max())
    tol = 1e-6 * max(1.0, longest ** 0.5) * 8
    assert (got.double() - want0).abs().max().item() <= tol * max(1.0, want0.abs().max().item())
    assert torch.equal(got, a) and torch.equal(got, again)
    assert (b.double() - want1).abs().max().item() <= tol * max(1.0, want1.abs().max().item())


def test_bounds_from_the_producer_kernels_change_no_bit(dev, monkeypatch):
    """f16x3 GEMM operand bounds emitted by seq_assemble_fwd / concat3_fwd / rgcn_bwd_prep (per-workgroup maxima of what
    they write) against RENET_FUSED_BOUNDS=0 (a renet_maxabs_partials pass per tensor): the maximum is the same number
    either way, so loss and every gradient are bit-identical -- and the passes over X, Xr, feat x2 and layer 2's g_loop
    are gone."""
    import model as M
    import ops
    import preprocess as P
    import renet_hip as K
    import synth
    if K.GEMM_MODE != 'f16x3':
        pytest.skip('bounds only exist in f16x3 mode')
    quads, num_ent, num_rels, _ = synth.make_stream('ICEWS18', seed=6, num_t=40)
    gd = P.build_graph_dict(quads, num_rels)
    hs, ho = P.HistoryIndex(quads, 's', 10), P.HistoryIndex(quads, 'o', 10)
    idx = np.random.RandomState(2).permutation(len(quads))[:256]
    calls = []
    real = K.maxabs_partials
    monkeypatch.setattr(K, 'maxabs_partials', lambda x: (calls.append(tuple(x.shape)), real(x))[1])
    res = []
    for fused in ('1', '0'):
        monkeypatch.setenv('RENET_FUSED_BOUNDS', fused)
        torch.manual_seed(11)
        ops.reset_seed_counter()
        net = M.RENet(num_ent, 200, num_rels, dropout=0.5, seq_len=10, num_k=10)
        gen = torch.Generator().manual_seed(3)
        net.global_emb = {int(t): torch.randn(1, 1, 200, generator=gen) * 0.1 for t in gd}
        net.to(dev).train()
        prep = net.prepare_both(quads[idx], hs.take(idx), ho.take(idx), gd)
        del calls[:]
        loss = net.loss_prepared_both(prep)
        loss.backward()
        torch.cuda.synchronize()
        res.append((loss.item(), {k: p.grad.clone() for k, p in net.named_parameters()}, len(calls)))
    assert res[0][0] == res[1][0]
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k
    assert res[0][2] <= res[1][2] - 5, (res[0][2], res[1][2])       # X, Xr, feat x2, g_loop no longer measured
