"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Plain-torch (CPU) stand-ins for the device wrappers of re-net_amd/renet_hip.py, each restating the contract of
its C-ABI entry point in include/renet_hip.h.  Purpose: tests/test_reference_drivers.py executes the reference's
UNMODIFIED drivers (pretrain.py / train.py / test.py) over this repository's API mirror in the build container,
which has no GPU -- argument parsing, pickle loading, the model's host logic, checkpoint writing and re-loading
all run for real, only the kernels are emulated.  install() patches the attributes of the imported `renet_hip`
module and returns an undo callable; nothing in re-net_amd/ knows about this file, and without install() the
product raises as usual when it is handed a non-HIP tensor.  Dropout must be 0 (the counter-based masks of the
kernels are not restated here).
"""
import numpy as np
import torch


def _no_drop(p):
    if float(p) != 0.0:
        raise NotImplementedError('the CPU emulation runs with dropout 0 only')


def gather_rows(table, idx, out=None):
    r = table[idx.long()]
    if out is not None:
        out.copy_(r)
        return out
    return r


def segment_add(src, plan, dst):
    for u in range(plan.num_segments):
        rows = plan.order[int(plan.seg_ptr[u]):int(plan.seg_ptr[u + 1])].long()
        dst[int(plan.target[u])] += src[rows].sum(dim=0)
    return dst


def segment_add2(src0, src1, plan, dst0, dst1):
    segment_add(src0, plan, dst0)
    segment_add(src1, plan, dst1)


def rgcn_gather_items_table(table, g, weight, type_shift, addend_table, drop_p, seed, relu, out, table16=None,
                            w16=None):
    ent = g.node_ent.long()
    return rgcn_gather_items(table[ent], g, weight, type_shift, False, addend_table[ent] if addend_table is not None
                             else None, drop_p, seed, relu, out, use_norm=True)


def compose_table_items(g):
    ent = g.node_ent.long()
    t = g.it_type.long()
    s_ = g.it_src.long()
    it_src_t = torch.where(t >= 0, ent[s_], s_).int()
    it_type_t = torch.where(t == -1, -3 - ent[s_], t).int()
    return it_src_t, it_type_t, ent[g.col.long()].int(), ent[g.e_src.long()].int()


def _blockmul(x, w, d, tr):
    si = d // 100
    xs = x.reshape(-1, 100, si)
    ws = w.reshape(-1, 100, si, si)
    y = torch.einsum('ebj,ebij->ebi', xs, ws) if tr else torch.einsum('ebi,ebij->ebj', xs, ws)
    return y.reshape(-1, d)


def rgcn_gather_items(x, g, weight, type_shift, transpose_w, addend, drop_p, seed, relu, out, use_norm=True,
                      pruned=False, src_limit=0, addend_rows=0, w16=None):
    _no_drop(drop_p)
    d, n_rows, T = x.shape[1], out.shape[0], weight.shape[0]
    rp = g.row_ptr.long()
    dst = torch.repeat_interleave(torch.arange(len(rp) - 1), rp[1:] - rp[:-1])
    src, et = g.col.long(), (g.etype.long() + type_shift) % T
    keep = dst < n_rows
    if src_limit:
        keep &= src < src_limit
    src, et, dst = src[keep], et[keep], dst[keep]
    ad = addend.clone() if addend is not None else None
    acc = torch.zeros(n_rows, d)
    if len(src):
        acc.index_add_(0, dst, _blockmul(x[src], weight[et], d, bool(transpose_w)))
    if use_norm:
        acc *= g.norm[:n_rows].view(-1, 1)
    if ad is not None:
        m = addend_rows if addend_rows else n_rows
        acc[:m] += ad[:m]
    out.copy_(torch.relu(acc) if relu else acc)
    return out


def rgcn_bwd_prep(g_out, out, norm, relu, drop_p, seed, gn, g_loop):
    _no_drop(drop_p)
    g = g_out * (out > 0) if relu else g_out
    gn.copy_(g * norm[:g.shape[0]].view(-1, 1))
    g_loop.copy_(g)


def rgcn_bwd_w(x, gn, e_src, e_dst, chunk_ptr, chunk_type, n_chunks, type_chunk_ptr, num_types, type_shift, dW,
               beta=0.0):
    d = x.shape[1]
    si = d // 100
    res = torch.zeros(num_types, 100, si, si)
    tcp, cp = type_chunk_ptr.tolist(), chunk_ptr.tolist()
    for t in range(num_types):
        if tcp[t + 1] > tcp[t]:
            e0, e1 = cp[tcp[t]], cp[tcp[t + 1]]
            xs = x[e_src[e0:e1].long()].reshape(-1, 100, si)
            gs = gn[e_dst[e0:e1].long()].reshape(-1, 100, si)
            res[(t + type_shift) % num_types] = torch.einsum('ebi,ebj->bij', xs, gs)
    res = res.reshape(num_types, -1)
    dW.copy_(res + beta * dW if beta else res)
    return dW


def gemm(a, b, ta=False, tb=False, out=None, bias=None, alpha=1.0, beta=0.0, split_k=None, mode=None):
    r = alpha * ((a.t() if ta else a) @ (b.t() if tb else b))
    if bias is not None:
        r = r + bias
    if out is None:
        return r
    out.copy_(r + beta * out if beta else r)
    return out


def colsum(x, out=None, beta=0.0):
    r = x.sum(dim=0)
    if out is None:
        return r
    out.copy_(r + beta * out if beta else r)
    return out


def scale_by_device_scalar(x, g):
    x.mul_(g.reshape(-1)[0])
    return x


def seq_assemble_fwd(h2, ent, rel, glob, subj_row, row_ent, row_rel, glob_row, drop_p, seed_x, seed_xr):
    _no_drop(drop_p)
    a, b, c, e = h2[subj_row.long()], ent[row_ent.long()], rel[row_rel.long()], glob[glob_row.long()]
    return torch.cat((a, b, c, e), dim=1), torch.cat((a, b, e), dim=1)


def seq_assemble_bwd(dx, dxr, step_off, num_steps, num_seq, d, drop_p, seed_x, seed_xr):
    _no_drop(drop_p)
    d_rows = dx[:, :d] + dxr[:, :d]
    d_ent = torch.zeros(num_seq, d)
    d_rel = torch.zeros(num_seq, d)
    off = step_off.tolist()
    for j in range(num_steps):
        n = off[j + 1] - off[j]
        d_ent[:n] += dx[off[j]:off[j + 1], d:2 * d] + dxr[off[j]:off[j + 1], d:2 * d]
        d_rel[:n] += dx[off[j]:off[j + 1], 2 * d:3 * d]
    return d_rows, d_ent, d_rel


def _gru_fwd_one(gi, off, hdim, w_hh, b_hh, out_rows):
    L = len(off) - 1
    b = off[1] - off[0] if L > 0 else 0
    out_rows = max(int(out_rows), b)
    h = torch.zeros(out_rows, hdim)
    saved = torch.zeros(gi.shape[0], 5 * hdim)
    for j in range(L):
        n = off[j + 1] - off[j]
        hp = h[:n].clone()
        gh = hp @ w_hh.t() + b_hh
        g = gi[off[j]:off[j + 1]]
        r = torch.sigmoid(g[:, :hdim] + gh[:, :hdim])
        z = torch.sigmoid(g[:, hdim:2 * hdim] + gh[:, hdim:2 * hdim])
        hn = gh[:, 2 * hdim:]
        nn_ = torch.tanh(g[:, 2 * hdim:] + r * hn)
        h[:n] = (1 - z) * nn_ + z * hp
        saved[off[j]:off[j + 1]] = torch.cat((r, z, nn_, hn, hp), dim=1)
    return h, saved


def _gru_bwd_one(dh_last, off, hdim, w_hh, saved):
    L = len(off) - 1
    s = saved.shape[0]
    d_gi, d_gh = torch.zeros(s, 3 * hdim), torch.zeros(s, 3 * hdim)
    dh = dh_last.clone()
    for j in range(L - 1, -1, -1):
        n = off[j + 1] - off[j]
        sv = saved[off[j]:off[j + 1]]
        r, z, nn_, hn, hp = (sv[:, k * hdim:(k + 1) * hdim] for k in range(5))
        g = dh[:n]
        dan = g * (1 - z) * (1 - nn_ * nn_)
        daz = g * (hp - nn_) * z * (1 - z)
        dar = dan * hn * r * (1 - r)
        d_gi[off[j]:off[j + 1]] = torch.cat((dar, daz, dan), dim=1)
        gh = torch.cat((dar, daz, dan * r), dim=1)
        d_gh[off[j]:off[j + 1]] = gh
        dh[:n] = g * z + gh @ w_hh
    return d_gi, d_gh


def _off(step_off_host):
    return [int(v) for v in step_off_host]


def gru_fwd(gi, step_off_host, hdim, w_hh, b_hh, out_rows=0):
    return _gru_fwd_one(gi, _off(step_off_host), hdim, w_hh, b_hh, out_rows)


def gru_bwd(dh_last, step_off_host, hdim, w_hh, saved):
    return _gru_bwd_one(dh_last, _off(step_off_host), hdim, w_hh, saved)


def gru_fwd_multi(gis, step_off_host, hdim, w_hhs, b_hhs, out_rows=0):
    res = [_gru_fwd_one(g, _off(step_off_host), hdim, w, b, out_rows) for g, w, b in zip(gis, w_hhs, b_hhs)]
    return [r[0] for r in res], [r[1] for r in res]


def gru_bwd_multi(dh_lasts, step_off_host, hdim, w_hhs, saveds):
    res = [_gru_bwd_one(d, _off(step_off_host), hdim, w, s) for d, w, s in zip(dh_lasts, w_hhs, saveds)]
    return [r[0] for r in res], [r[1] for r in res]


def gru_fwd_layouts(gis, step_offs, hdim, w_hhs, b_hhs, out_rows):
    res = [_gru_fwd_one(g, _off(o), hdim, w, b, r) for g, o, w, b, r in zip(gis, step_offs, w_hhs, b_hhs, out_rows)]
    return [r[0] for r in res], [r[1] for r in res]


def gru_bwd_layouts(dh_lasts, step_offs, hdim, w_hhs, saveds, out_bf16=False):
    assert not out_bf16, 'the CPU emulation has no bf16-storage mode'
    res = [_gru_bwd_one(d, _off(o), hdim, w, s) for d, o, w, s in zip(dh_lasts, step_offs, w_hhs, saveds)]
    return [r[0] for r in res], [r[1] for r in res]


def concat3_fwd(a, ia, hmid, c, ic, drop_p, seed):
    _no_drop(drop_p)
    parts = [a[ia.long()], hmid] + ([c[ic.long()]] if c is not None else [])
    return torch.cat(parts, dim=1)


def concat3_bwd(dfeat, d, parts, drop_p, seed):
    _no_drop(drop_p)
    return (dfeat[:, :d].contiguous(), dfeat[:, d:2 * d].contiguous(),
            dfeat[:, 2 * d:3 * d].contiguous() if parts == 3 else None)


def dropout(x, drop_p, seed):
    _no_drop(drop_p)
    return x.clone()


def softmax_ce(logits, target, grad_scale, want_grad, row_loss=None):
    lse = torch.logsumexp(logits, dim=1)
    rows = torch.arange(logits.shape[0])
    loss = lse - logits[rows, target.long()]
    if want_grad:
        g = torch.softmax(logits, dim=1)
        g[rows, target.long()] -= 1.0
        logits.copy_(g * grad_scale)
    if row_loss is not None:
        row_loss.copy_(loss)
        return row_loss
    return loss


def segment_pool_fwd(h, seg_ptr, num_graphs, is_max):
    d = h.shape[1]
    out = torch.zeros(num_graphs, d)
    arg = torch.zeros(num_graphs, d, dtype=torch.int32)
    sp = seg_ptr.tolist()
    for gi in range(num_graphs):
        rows = h[sp[gi]:sp[gi + 1]]
        if is_max:
            v, a = rows.max(dim=0)
            out[gi], arg[gi] = v, (a + sp[gi]).int()
        else:
            out[gi] = rows.mean(dim=0)
    return out, arg


def segment_pool_bwd(dout, seg_ptr, arg, num_graphs, is_max, n):
    d = dout.shape[1]
    dh = torch.zeros(n, d)
    sp = seg_ptr.tolist()
    cols = torch.arange(d)
    for gi in range(num_graphs):
        if is_max:
            dh[arg[gi].long(), cols] += dout[gi]
        else:
            dh[sp[gi]:sp[gi + 1]] += dout[gi] / max(sp[gi + 1] - sp[gi], 1)
    return dh


EMULATED = ['gather_rows', 'segment_add', 'segment_add2', 'rgcn_gather_items', 'rgcn_gather_items_table',
            'compose_table_items', 'rgcn_bwd_prep', 'rgcn_bwd_w', 'gemm', 'colsum',
            'scale_by_device_scalar', 'seq_assemble_fwd', 'seq_assemble_bwd', 'gru_fwd', 'gru_bwd', 'gru_fwd_multi',
            'gru_bwd_multi', 'gru_fwd_layouts', 'gru_bwd_layouts', 'concat3_fwd', 'concat3_bwd', 'dropout', 'softmax_ce', 'segment_pool_fwd',
            'segment_pool_bwd']


def install():
    """Patch the imported renet_hip module; returns undo()."""
    import renet_hip as K
    saved = {n: getattr(K, n) for n in EMULATED}
    g = globals()
    for n in EMULATED:
        setattr(K, n, g[n])

    def undo():
        for n, f in saved.items():
            setattr(K, n, f)
    return undo
