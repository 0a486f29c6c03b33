"""GPU tests of the item-stream gather-SpMM (renet_rgcn_gather_items; RGCN.py:79-94 + 42-50) through the C ABI:
against an fp64 numpy evaluation of the operator's definition, against the plain-CSR kernel, for every n_hidden,
both block orientations, the fused epilogue (norm, self-loop addend with dropout mask, ReLU), the pruned launches
(row prefix / source limit / addend prefix), degenerate graphs, several plan parameters and every UNR variant."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 're-net_amd')


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need a HIP device'
    import renet_hip
    renet_hip.lib()
    return torch.device('cuda:0')


def random_graph(seed, n, avg_deg, num_types, hub=0, zero=10):
    rng = np.random.RandomState(seed)
    deg = np.minimum(rng.zipf(1.7, n), 60) if avg_deg is None else rng.poisson(avg_deg, n)
    if hub:
        deg[rng.randint(0, n, 6)] = rng.randint(hub // 2, hub, 6)        # hub rows, several index windows long
    if zero:
        deg[rng.randint(0, n, zero)] = 0
    dst = np.repeat(np.arange(n), deg)
    src = rng.randint(0, n, len(dst))
    et = rng.randint(0, num_types, len(dst))
    return src, dst, et


def reference(x, src, dst, et, w, d, T, shift, tr, norm, addend, relu, n_out, src_limit, addend_rows):
    """The operator's definition in fp64 (include/renet_hip.h)."""
    si = d // 100
    xs = x.astype(np.float64)[src].reshape(-1, 100, si)
    wt = w.astype(np.float64)[(et + shift) % T].reshape(-1, 100, si, si)
    msg = np.einsum('ebj,ebij->ebi', xs, wt) if tr else np.einsum('ebi,ebij->ebj', xs, wt)
    msg = msg.reshape(-1, d)
    keep = (dst < n_out) & ((src < src_limit) if src_limit else True)
    out = np.zeros((n_out, d))
    np.add.at(out, dst[keep], msg[keep])
    if norm is not None:
        out *= norm[:n_out, None]
    if addend is not None:
        m = addend_rows if addend_rows else n_out
        out[:m] += addend[:m]
    return np.maximum(out, 0) if relu else out


@pytest.mark.parametrize('d', [100, 200, 400])
@pytest.mark.parametrize('heavy,budget', [(24, 32), (4, 8), (62, 1)])
def test_item_gather_matches_fp64_definition(dev, d, heavy, budget):
    import graph as G
    import renet_hip as K
    T, n = 14, 3000
    src, dst, et = random_graph(d + heavy, n, None, T, hub=300)
    hb = G.HostBatch().set_edges(n, src, dst, et, T, heavy=heavy)
    n_out = 1100
    hb.set_out_rows(n_out, src, dst, et)
    hb.set_gather_plan(n_out, heavy=heavy, budget=budget)
    g = G.DeviceGraph(hb, dev)
    assert g.heavy_rows is not None and int(np.diff(hb.row_ptr).max()) > 128
    rng = np.random.RandomState(1)
    x = rng.randn(n, d).astype(np.float32)
    w = (rng.randn(T, d * d // 100) * 0.3).astype(np.float32)
    ad = rng.randn(n, d).astype(np.float32)
    tx, tw, tad = (torch.from_numpy(a).to(dev) for a in (x, w, ad))
    for tr in (False, True):
        for shift in (0, T // 2):
            # full launch, fused epilogue
            out = torch.empty(n, d, device=dev)
            K.rgcn_gather_items(tx, g, tw, shift, tr, tad, 0.0, 0, not tr, out, use_norm=not tr)
            ref = reference(x, src, dst, et, w, d, T, shift, tr, None if tr else hb.norm, ad, not tr, n, 0, 0)
            np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-4, atol=1e-4)
            # the plain-CSR kernel computes the same thing
            out2 = torch.empty(n, d, device=dev)
            K.rgcn_gather(tx, g.row_ptr, g.col, g.etype, None if tr else g.norm, tw, shift, tr, tad, 0.0, 0, not tr,
                          out2, g.heavy_rows, g.heavy_thresh)
            np.testing.assert_allclose(out2.cpu().numpy(), ref, rtol=1e-4, atol=1e-4)
    # pruned forward: rows [0, n_out) only, no addend, in place of nothing
    outp = torch.empty(n_out, d, device=dev)
    K.rgcn_gather_items(tx, g, tw, 0, False, None, 0.0, 0, False, outp, use_norm=True, pruned=True)
    refp = reference(x, src, dst, et, w, d, T, 0, False, hb.norm, None, False, n_out, 0, 0)
    np.testing.assert_allclose(outp.cpu().numpy(), refp, rtol=1e-4, atol=1e-4)
    # pruned backward: all rows are outputs, sources >= n_out skipped, in-place addend on the row prefix only
    gn = rng.randn(n_out, d).astype(np.float32)
    dh0 = np.zeros((n, d), np.float32)
    dh0[:n_out] = rng.randn(n_out, d)
    dh = torch.from_numpy(dh0.copy()).to(dev)
    dh[n_out:] = float('nan')                      # rows >= addend_rows must be overwritten, never read
    K.rgcn_gather_items(torch.from_numpy(gn).to(dev), g, tw, T // 2, True, dh, 0.0, 0, False, dh, use_norm=False,
                        pruned=True, src_limit=n_out, addend_rows=n_out)
    gn_full = np.zeros((n, d), np.float32)
    gn_full[:n_out] = gn
    refb = reference(gn_full, src, dst, et, w, d, T, T // 2, True, None, dh0, False, n, n_out, n_out)
    np.testing.assert_allclose(dh.cpu().numpy(), refb, rtol=1e-4, atol=1e-4)
    # run-to-run bit reproducibility
    o1, o2 = torch.empty(n, d, device=dev), torch.empty(n, d, device=dev)
    K.rgcn_gather_items(tx, g, tw, 0, False, tad, 0.0, 0, True, o1)
    K.rgcn_gather_items(tx, g, tw, 0, False, tad, 0.0, 0, True, o2)
    assert torch.equal(o1, o2)


def test_item_gather_dropout_mask_equals_the_csr_kernels(dev):
    """The fused self-loop dropout mask is keyed by (seed, row, chunk): both kernels and the backward prologue
    regenerate the same one."""
    import graph as G
    import renet_hip as K
    T, n, d = 10, 2000, 200
    src, dst, et = random_graph(5, n, 3.0, T, hub=100)
    g = G.DeviceGraph(G.HostBatch.from_edges(n, src, dst, et, T // 2), dev)
    torch.manual_seed(0)
    x, ad = torch.randn(n, d, device=dev), torch.randn(n, d, device=dev)
    w = torch.randn(T, 2 * d, device=dev) * 0.2
    a, b = torch.empty(n, d, device=dev), torch.empty(n, d, device=dev)
    K.rgcn_gather_items(x, g, w, 0, False, ad, 0.5, 1234567, True, a)
    K.rgcn_gather(x, g.row_ptr, g.col, g.etype, g.norm, w, 0, False, ad, 0.5, 1234567, True, b, g.heavy_rows,
                  g.heavy_thresh)
    np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-5, atol=1e-5)
    c = torch.empty(n, d, device=dev)
    K.rgcn_gather_items(x, g, w, 0, False, ad, 0.0, 0, True, c)
    assert float((a - c).abs().max()) > 0.1                      # the mask did something


def test_item_gather_degenerate_graphs(dev):
    import graph as G
    import renet_hip as K
    d, T = 200, 6
    w = torch.randn(T, 2 * d, device=dev)
    for n, src, dst, et in ((1, [], [], []), (5, [], [], []), (3, [0, 0, 0, 1], [2, 2, 2, 2], [0, 1, 2, 3]),
                            (2, [0] * 200, [1] * 200, [5] * 200)):
        hb = G.HostBatch.from_edges(n, np.asarray(src, np.int64), np.asarray(dst, np.int64),
                                    np.asarray(et, np.int64), T // 2)
        g = G.DeviceGraph(hb, dev)
        x = torch.randn(n, d, device=dev)
        ad = torch.randn(n, d, device=dev)
        out = torch.full((n, d), float('nan'), device=dev)
        K.rgcn_gather_items(x, g, w, 0, False, ad, 0.0, 0, False, out)
        ref = reference(x.cpu().numpy(), np.asarray(src, np.int64), np.asarray(dst, np.int64),
                        np.asarray(et, np.int64), w.cpu().numpy(), d, T, 0, False, hb.norm, ad.cpu().numpy(),
                        False, n, 0, 0)
        np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-4, atol=1e-4)


_UNR_CHECK = r'''
import sys
import numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import graph as G, renet_hip as K
from test_gpu_gather import random_graph, reference
dev = torch.device('cuda:0')
for d in (100, 200, 400):
    T, n = 12, 2500
    src, dst, et = random_graph(d, n, None, T, hub=260)
    hb = G.HostBatch.from_edges(n, src, dst, et, T // 2)
    g = G.DeviceGraph(hb, dev)
    rng = np.random.RandomState(2)
    x = rng.randn(n, d).astype(np.float32); w = (rng.randn(T, d * d // 100) * 0.3).astype(np.float32)
    ad = rng.randn(n, d).astype(np.float32)
    for tr in (False, True):
        out = torch.empty(n, d, device=dev)
        K.rgcn_gather_items(torch.from_numpy(x).to(dev), g, torch.from_numpy(w).to(dev), 3, tr, torch.from_numpy(ad).to(dev),
                            0.0, 0, True, out, use_norm=True)
        ref = reference(x, src, dst, et, w, d, T, 3, tr, hb.norm, ad, True, n, 0, 0)
        np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-4, atol=1e-4)
print('ok')
'''


@pytest.mark.parametrize('unr', ['2', '3', '4', '6', '8'])
def test_item_gather_every_unroll_variant(dev, unr):
    """RENET_GATHER_UNR selects the number of items in flight per wave (read once per process)."""
    env = dict(os.environ, RENET_GATHER_UNR=unr)
    r = subprocess.run([sys.executable, '-c', _UNR_CHECK, PKG, os.path.dirname(os.path.abspath(__file__))], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'ok' in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


# ---------------------------------------------------------------------------------------------
# first layer addressed through the entity table (ops.RGCNTableLayerFn, renet_rgcn_gather_items_table):
# == the layer on the materialised h0 = table[node_ent] (ops.GatherRowsFn + ops.RGCNLayerFn), forward and every
# gradient, with dropout on (the mask is keyed by (node row, column) in both forms) and hub rows present
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('d', [100, 200, 400])
@pytest.mark.parametrize('drop_p', [0.0, 0.4])
def test_table_addressed_first_layer_equals_materialised_layer(dev, d, drop_p):
    import graph as G
    import ops
    n, n_ent, num_rels = 3000, 700, 9
    T = 2 * num_rels
    rng = np.random.RandomState(17 + d)
    # paired edges (both directions of every fact, utils.py:74-76) so that the transposed gather of the backward
    # pass is the same CSR
    m = 6000
    a, b, r = rng.randint(0, n, m), rng.randint(0, n, m), rng.randint(0, num_rels, m)
    hubs = rng.randint(0, n, 4)
    a[:400] = hubs[rng.randint(0, 4, 400)]                                   # a few rows with ~100 in/out edges
    src, dst, et = np.concatenate((a, b)), np.concatenate((b, a)), np.concatenate((r, r + num_rels))
    hb = G.HostBatch.from_edges(n, src, dst, et, num_rels)
    hb.node_ent = rng.randint(0, n_ent, n).astype(np.int32)
    hb.plan_node_ent = G.SegPlan.host(hb.node_ent)
    g = G.DeviceGraph(hb, dev)
    assert g.heavy_rows is not None and g.heavy_rows.numel() >= 2

    def leaf(a_):
        return torch.from_numpy(a_.astype(np.float32)).to(dev).requires_grad_(True)
    tab0, w0, l0 = rng.randn(n_ent, d) * 0.3, rng.randn(T, d * d // 100) * 0.2, rng.randn(d, d) * 0.1
    cot = torch.from_numpy(rng.randn(n, d).astype(np.float32)).to(dev)
    res = []
    # pre-activations within fp32 rounding of zero take either side of the ReLU depending on the summation order
    # (one flipped element moves a gradient entry by ~1e-2 of its tensor's max): give those elements no cotangent
    with torch.no_grad():
        pre = ops.RGCNTableLayerFn.apply(leaf(tab0).detach(), leaf(w0).detach(), leaf(l0).detach(), g, False, False,
                                         drop_p, 1234)
        cot = cot * (pre.abs() > 1e-4)
    for table_path in (True, False):
        tab, w, lw = leaf(tab0), leaf(w0), leaf(l0)
        if table_path:
            out = ops.RGCNTableLayerFn.apply(tab, w, lw, g, False, True, drop_p, 1234)
        else:
            h0 = ops.GatherRowsFn.apply(tab, g.node_ent, g.plan_node_ent)
            out = ops.RGCNLayerFn.apply(h0, w, lw, g, False, True, drop_p, 1234, None)
        (out * cot).sum().backward()
        torch.cuda.synchronize()
        res.append([t.detach().cpu().double().numpy() for t in (out, tab.grad, w.grad, lw.grad)])
    for name, x, y in zip(('out', 'd_table', 'd_weight', 'd_loop'), res[0], res[1]):
        scale = np.abs(y).max()
        assert scale > 0
        assert np.abs(x - y).max() <= (1e-6 if name == 'out' else 2e-5) * scale, (name, np.abs(x - y).max(), scale)
    # and against the fp64 definition (dropout off)
    if drop_p == 0.0:
        norm = hb.norm
        h0 = tab0[hb.node_ent]
        want = reference(h0, src, dst, et, w0, d, T, 0, False, norm, h0 @ l0, True, n, 0, 0)
        assert np.abs(res[0][0] - want).max() <= 2e-5 * np.abs(want).max()


@pytest.mark.gpu
@pytest.mark.parametrize('d', [100, 200, 400])
def test_bf16_stored_operands_equal_fp32_kernels_on_rounded_operands(dev, d):
    """renet_rgcn_gather_items_bf16 / _table_bf16 widen bf16 relation blocks (and table rows) in registers and run the
    same fp32 arithmetic in the same order: BIT-identical to the fp32 kernels fed the bf16-rounded values."""
    import graph as G
    import renet_hip as K
    n, n_ent, num_rels = 3000, 700, 9
    T = 2 * num_rels
    rng = np.random.RandomState(5 + d)
    m = 6000
    a, b, r = rng.randint(0, n, m), rng.randint(0, n, m), rng.randint(0, num_rels, m)
    hubs = rng.randint(0, n, 4)
    a[:400] = hubs[rng.randint(0, 4, 400)]
    src, dst, et = np.concatenate((a, b)), np.concatenate((b, a)), np.concatenate((r, r + num_rels))
    hb = G.HostBatch.from_edges(n, src, dst, et, num_rels)
    hb.node_ent = rng.randint(0, n_ent, n).astype(np.int32)
    hb.plan_node_ent = G.SegPlan.host(hb.node_ent)
    g = G.DeviceGraph(hb, dev)
    assert g.heavy_rows is not None and g.heavy_rows.numel() >= 2

    def t32(a_):
        return torch.from_numpy(a_.astype(np.float32)).to(dev)
    x, w, tab = t32(rng.randn(n, d) * 0.3), t32(rng.randn(T, d * d // 100) * 0.2), t32(rng.randn(n_ent, d) * 0.3)
    add, add_tab = t32(rng.randn(n, d)), t32(rng.randn(n_ent, d))
    w16, tab16 = K.pack_bf16(w), K.pack_bf16(tab)
    w_r, tab_r = w.bfloat16().float(), tab.bfloat16().float()
    for tr in (False, True):
        for use_add in (True, False):
            got = torch.empty(n, d, device=dev)
            want = torch.empty(n, d, device=dev)
            ad = add if use_add else None
            K.rgcn_gather_items(x, g, w, 0 if not tr else num_rels, tr, ad, 0.0, 0, not tr, got, use_norm=not tr,
                                w16=w16)
            K.rgcn_gather_items(x, g, w_r, 0 if not tr else num_rels, tr, ad, 0.0, 0, not tr, want, use_norm=not tr)
            torch.cuda.synchronize()
            assert torch.equal(got, want), (d, tr, use_add, (got - want).abs().max().item())
    for drop_p in (0.0, 0.3):
        got = torch.empty(n, d, device=dev)
        want = torch.empty(n, d, device=dev)
        K.rgcn_gather_items_table(tab, g, w, 0, add_tab, drop_p, 77, True, got, table16=tab16, w16=w16)
        K.rgcn_gather_items_table(tab_r, g, w_r, 0, add_tab, drop_p, 77, True, want)
        torch.cuda.synchronize()
        assert torch.equal(got, want), (d, drop_p, (got - want).abs().max().item())
