"""torch.autograd glue around the C-ABI kernels (renet_hip.py).  Each Function's forward AND backward
run hand-written HIP kernels; torch only owns the tensors and wires the graph.  No CPU / eager
fallback exists: on a non-HIP tensor these raise."""
import contextlib
import threading
import functools
import ctypes
import os

import numpy as np
import torch
from torch.autograd import Function

import renet_hip as K

_seed_state = {'counter': 0}

# A second HIP stream inside ONE autograd Function: independent kernel chains of a Function (the relation head next
# to the entity head, encoder_r's GEMMs next to encoder's, a bandwidth-bound bias column sum next to the matrix-bound
# GEMMs that read the same gradient) are enqueued on two streams between a fork and a join, so that the tail wave of
# one launch and the few-workgroup kernels of the small chain fill the CUs the other leaves idle.  Fork and join both
# lie inside the Function: everything outside sees plain stream order.  Memory: a tensor allocated under the side
# stream is only ever written there after a later fork (which orders the side stream behind every main-stream
# consumer enqueued so far), and the inputs the side chain reads stay referenced by the caller until after the join.
# RENET_SIDE_STREAM=0, or kernel timing (K.set_timer: per-kernel events want serial launches), keeps one stream.
SIDE_STREAM = os.environ.get('RENET_SIDE_STREAM', '1') != '0'
_side_streams = {}


class _Side(object):
    def __init__(self, device):
        self.on = SIDE_STREAM and device.type == 'cuda' and K._timer is None
        if self.on:
            key = device.index if device.index is not None else torch.cuda.current_device()
            if key not in _side_streams:
                _side_streams[key] = torch.cuda.Stream(device=device)
            self.main = torch.cuda.current_stream(device)
            self.side = _side_streams[key]
            self.side.wait_stream(self.main)                       # fork

    def __call__(self):
        return torch.cuda.stream(self.side) if self.on else contextlib.nullcontext()

    def join(self):
        if self.on:
            self.main.wait_stream(self.side)


# Deferred joins (round 4).  Some side-stream work has no consumer inside the backward pass at all -- the weight / bias
# gradients of the GRU input projections and recurrences feed nothing but the optimizer -- so its join can wait until
# the optimizer step instead of the end of the Function: the main stream goes on into sequence-assembly and RGCN
# backward (~0.5 ms of small, latency-bound kernels) while the side stream runs the dW GEMMs.  Only a caller that
# PROMISES the final join may turn this on: parallel.HipAdam's step_scope sets `DEFER_WEIGHT_GRADS` for the step and
# HipAdam.step() calls join_deferred() before it reads a gradient.  Outside such a scope (train.py's own loop with
# clip_grad_norm_ + torch Adam) every Function joins before it returns, as before.  Tensors the deferred kernels read
# are handed to the allocator with record_stream, so that freeing them on the main stream cannot recycle their memory
# under the side stream -- and EVERYTHING the deferred kernels read (those tensors, the batch graph with its index arrays and
# item plans, operand shells with their bound partials) is also kept alive by a strong reference in `_deferred` until
# join_deferred() has made the main stream wait: a caller that drops `loss` or the prepared batch between backward() and
# step() cannot let the allocator recycle memory the side stream is still reading.
DEFER_WEIGHT_GRADS = False
_deferred = []             # (main stream, side stream, keep-alive objects) of un-joined side work


def _deferred_scope(device, tensors, keep=()):
    """-> a _Side whose stream takes gradient-only work that stays UN-JOINED until join_deferred() (None when deferral
    is off or no side stream exists).  `tensors`: what that work reads and the caller frees afterwards; `keep`: further
    objects (the DeviceGraph, operand shells) whose device memory the work reads."""
    if not DEFER_WEIGHT_GRADS:
        return None
    sd = _Side(device)
    if not sd.on:
        return None
    for t in tensors:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            t.record_stream(sd.side)
    _deferred.append((sd.main, sd.side, (list(tensors), list(keep))))
    return sd


def join_deferred():
    """Make every stream that forked deferred work wait for it (no-op if nothing is pending); the kept objects are
    released after the wait has been enqueued (their memory returns to the allocator on the main stream, behind it)."""
    while _deferred:
        main, side, keep = _deferred.pop()
        main.wait_stream(side)
        del keep


def _rank():
    import torch.distributed as dist
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


# Exact data-parallel split of ONE batch (graph.shard_sequences, bench --scaling exact): every rank evaluates the SAME
# batch graph, so the dropout masks of the graph-side sites (the RGCN layers' self-loop dropout) must be the same on
# every rank for the N-rank step to equal the 1-rank step; the per-sequence sites (sequence assembly, score heads) act
# on rows only this rank owns and keep rank-dependent seeds.  Set by the caller that shards a batch that way.
SHARED_GRAPH_SEEDS = False       # process default; a sharded batch sets it for ITS forward pass only (shared_graph_seeds)
_shared_tls = threading.local()  # .flag: the scope's value on THIS thread (seeds are drawn in forward, on the caller's thread)


@contextlib.contextmanager
def shared_graph_seeds(flag):
    """Forward pass of a batch whose GRAPH is replicated on every rank (RENet.loss_prepared_both on a batch prepared
    with shard=(rank, world)): the graph-side dropout sites draw rank-independent seeds inside this scope.  Seeds are
    only drawn in forward (the backward pass replays them from ctx), so a scope around the forward call suffices; the
    flag travels with the prepared batch instead of a module global that a caller has to set per step (review r3)."""
    old = getattr(_shared_tls, 'flag', None)
    _shared_tls.flag = bool(flag)
    try:
        yield
    finally:
        _shared_tls.flag = old


def next_seed(graph_site=False):
    """Fresh 63-bit seed for one dropout site, derived from torch's global seed (train.py:31 seeds it), the
    data-parallel rank (every rank runs with the same torch seed, and identical masks on every rank would
    correlate the ranks' dropout noise) and a per-process site counter.  graph_site: a site on the batch GRAPH --
    rank-independent when the ranks replicate one graph (SHARED_GRAPH_SEEDS)."""
    _seed_state['counter'] += 1
    scoped = getattr(_shared_tls, 'flag', None)
    shared = SHARED_GRAPH_SEEDS if scoped is None else scoped
    rank = 0 if (graph_site and shared) else _rank()
    x = torch.initial_seed() * 0x9E3779B1 + (rank + 1) * 0xC2B2AE3D27D4EB4F + _seed_state['counter'] * 0x85EBCA77
    return x & 0x7FFFFFFFFFFFFFFF


def reset_seed_counter(value=0):
    """Restart the per-process dropout site counter (reproducible runs: call next to torch.manual_seed)."""
    _seed_state['counter'] = int(value)


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# When a parameter already owns a persistent .grad buffer (parallel.FlatGrads / zero_grad(set_to_none=False)),
# the backward kernels accumulate STRAIGHT into it (GEMM beta=1, segmented scatter-add) and return None for
# that input, instead of materialising a full-size gradient tensor that autograd then adds: for ent_embeds
# (4 uses per direction, 18 MB each) and linear.weight (55 MB) that removes ~45 add/fill kernels per step.
INPLACE_GRADS = True

# Registered by parallel.HipAdam when gradients are exchanged between ranks: callbacks keyed by PARAMETER identity,
# called with the parameter right after a Function accumulated its contribution into param.grad in place (the score
# head's weight and bias), so that the all-reduce of that bucket can start while the rest of the backward pass runs.
# A registry, not one global slot: two optimizers (RENet + RENet_global, tests) each keep their own hooks, and
# unregister_grad_done_hook releases everything the callback keeps alive.
_grad_done_hooks = {}          # id(param) -> {token: callback}
_hook_tokens = {'next': 0}


_hook_wanted = {}              # token -> predicate: does this hook want notifications right now (None: always)


def register_grad_done_hook(params, callback, wanted=None):
    """wanted: optional callable() -> bool; callers that pay for a notification (events, a hook stream: step_plan.StepFn) ask
    hooks_wanted(param) first -- the reducer of a process without a process group wants none."""
    _hook_tokens['next'] += 1
    token = _hook_tokens['next']
    for p in params:
        _grad_done_hooks.setdefault(id(p), {})[token] = callback
    _hook_wanted[token] = wanted
    return token


def hooks_wanted(param):
    cbs = _grad_done_hooks.get(id(param))
    if not cbs:
        return False
    for token in cbs:
        w = _hook_wanted.get(token)
        if w is None or w():
            return True
    return False


def unregister_grad_done_hook(token):
    _hook_wanted.pop(token, None)
    for pid in list(_grad_done_hooks):
        _grad_done_hooks[pid].pop(token, None)
        if not _grad_done_hooks[pid]:
            del _grad_done_hooks[pid]


def grad_done(param):
    cbs = _grad_done_hooks.get(id(param))
    if cbs:
        for cb in list(cbs.values()):
            cb(param)

# Test hook (tests/test_gpu_config.py): when set to a callable(name, tensor), the training path reports its
# internal activations (GRU final states, entity logits before the in-place CE) -- None in production.
debug_tap = None


def _fwd_mode(fwd):
    """Function.forward wrapper: remembers the GEMM mode the forward pass ran in (renet_hip.gemm_mode scopes are entered
    by the MODEL around its forward code; autograd runs backward later, on its own thread, outside any such scope)."""
    @functools.wraps(fwd)
    def forward(ctx, *a, **k):
        ctx.gemm_mode = K.current_mode()
        return fwd(ctx, *a, **k)
    return forward


def _bwd_mode(bwd):
    """Function.backward wrapper: runs the backward pass in the mode its forward pass ran in."""
    @functools.wraps(bwd)
    def backward(ctx, *g):
        with K.gemm_mode(getattr(ctx, 'gemm_mode', None)):
            return bwd(ctx, *g)
    return backward


def grad_target(t):
    """The slice of a leaf parameter's existing .grad that corresponds to tensor `t` (the parameter itself or
    a contiguous row-slice view of it), or None if there is nothing to accumulate into.
    Always called from BACKWARD (the Functions keep `t` on ctx, not the resolved buffer): a loop that calls
    zero_grad(set_to_none=True) or replaces .grad between forward and backward must not leave the kernels
    accumulating into an orphaned tensor -- with .grad gone the Functions fall back to returning a
    materialised gradient and autograd sets .grad as usual."""
    if not INPLACE_GRADS or t is None:
        return None
    base = t if t.is_leaf else getattr(t, '_base', None)
    if base is None or not base.is_leaf or not base.requires_grad or base.grad is None:
        return None
    if not base.grad.is_contiguous() or base.grad.shape != base.shape:
        return None
    if t is base:
        return base.grad
    if t.dim() == 2 and base.dim() == 2 and t.is_contiguous() and t.shape[1] == base.shape[1]:
        off = t.storage_offset() - base.storage_offset()
        if off >= 0 and off % base.shape[1] == 0:
            r0 = off // base.shape[1]
            return base.grad[r0:r0 + t.shape[0]]
    return None


class GatherRowsFn(Function):
    """out = table[idx]  (utils.py:239 h0 = ent_embeds[id]); backward = deterministic segmented add."""

    @staticmethod
    @_fwd_mode
    def forward(ctx, table, idx, plan):
        ctx.plan, ctx.shape, ctx.src = plan, table.shape, table
        return K.gather_rows(_c(table), idx)

    @staticmethod
    @_bwd_mode
    def backward(ctx, g):
        tgt = grad_target(ctx.src)
        if tgt is not None:
            K.segment_add(_c(g), ctx.plan, tgt)
            return None, None, None
        d = torch.zeros(ctx.shape, device=g.device, dtype=torch.float32)
        K.segment_add(_c(g), ctx.plan, d)
        return d, None, None


class RGCNLayerFn(Function):
    """One RGCNBlockLayer (RGCN.py:33-51,79-94): self-loop GEMM + fused gather-SpMM epilogue.
    n_out < N evaluates the layer only for the first n_out rows (the batch builder numbers the rows that
    are read afterwards -- the subject rows, Aggregator.py:139-140 -- first); exact for those rows."""

    @staticmethod
    @_fwd_mode
    def forward(ctx, h, weight, loop_weight, g, reverse, relu, drop_p, seed, n_out):
        ctx.src_w, ctx.src_loop = weight, loop_weight
        h, weight, loop_weight = _c(h), _c(weight), _c(loop_weight)
        n = h.shape[0]
        n_out = n if (n_out is None or n_out >= n) else int(n_out)
        pruned = n_out < n
        shift = g.num_types // 2 if reverse else 0                     # type_o = type_s +- R (utils.py:75-76)
        h_op = K.operand(h[:n_out])                                    # (bf16 mode: packed once, reused by backward)
        out = K.gemm(h_op, loop_weight)                                # RGCN.py:35
        K.rgcn_gather_items(h, g, weight, shift, False, out, drop_p, seed, relu, out, use_norm=True, pruned=pruned,
                            w16=K.gather_weight_bf16(weight))
        ctx.g, ctx.relu, ctx.drop_p, ctx.seed, ctx.shift, ctx.n_out = g, relu, drop_p, seed, shift, n_out
        ctx.h_op = h_op if K.is_handle(h_op) else None
        ctx.save_for_backward(h, weight, loop_weight, out)
        return out

    @staticmethod
    @_bwd_mode
    def backward(ctx, g_out):
        h, weight, loop_weight, out = ctx.saved_tensors
        g, n_out = ctx.g, ctx.n_out
        tgt_loop, tgt_w = grad_target(ctx.src_loop), grad_target(ctx.src_w)
        g_out = _c(g_out)
        n, d = h.shape
        pruned = n_out < n
        gn = torch.empty(n_out, d, device=h.device, dtype=torch.float32)
        g_loop = torch.empty(n_out, d, device=h.device, dtype=torch.float32)
        K.rgcn_bwd_prep(g_out, out, g.norm, ctx.relu, ctx.drop_p, ctx.seed, gn, g_loop)
        dh = torch.empty(n, d, device=h.device, dtype=torch.float32)
        # dh = sum over out-edges W[type]^T gn[dst]  == same CSR rows, the PAIRED edge's type; with a pruned
        # forward only destinations < n_out carry gradient: skip the other sources.  Launched right behind the
        # kernel that produced gn (still cache resident); the self-loop term is then ACCUMULATED by its GEMM
        # (beta = 1, rows < n_out): the matrix-bound GEMM hides the read of dh that the bandwidth-bound gather
        # would otherwise pay for as an addend (64 -> 76 us per launch on the merged batch).
        pair_shift = (ctx.shift + g.num_types // 2) % g.num_types
        K.rgcn_gather_items(gn, g, weight, pair_shift, True, None, 0.0, 0, False, dh, use_norm=False, pruned=pruned,
                            src_limit=n_out if pruned else 0, w16=K.gather_weight_bf16(weight))
        gl_op = K.operand(g_loop)
        h_op = ctx.h_op if ctx.h_op is not None else h[:n_out]
        K.gemm(gl_op, loop_weight, tb=True, out=dh[:n_out], beta=1.0)      # += g_loop @ W_loop^T (rows < n_out)
        acc = tgt_w is not None                                        # straight into weight.grad (beta = 1)
        d_w = tgt_w if acc else torch.empty_like(weight)

        def weight_grads():
            if pruned:
                K.rgcn_bwd_w(h, gn, g.e_src2, g.e_dst2, g.chunk_ptr2, g.chunk_type2, g.n_chunks2, g.type_chunk_ptr2,
                             g.num_types, ctx.shift, d_w, beta=1.0 if acc else 0.0)
            else:
                K.rgcn_bwd_w(h, gn, g.e_src, g.e_dst, g.chunk_ptr, g.chunk_type, g.n_chunks, g.type_chunk_ptr,
                             g.num_types, ctx.shift, d_w, beta=1.0 if acc else 0.0)
            if tgt_loop is not None:                                   # h^T @ g_loop (auto split-K), accumulated
                K.gemm(h_op, gl_op, ta=True, out=tgt_loop, beta=1.0)
                return None
            return K.gemm(h_op, gl_op, ta=True)
        # the relation-block and self-loop weight gradients feed only the optimizer, and nothing else writes their
        # buffers: inside a declared step they run on the side stream, un-joined, under the rest of the backward pass
        sd = _deferred_scope(h.device, (h, gn, g_loop, getattr(gl_op, 'p', None), getattr(gl_op, 'part', None),
                                        getattr(h_op, 'p', None), getattr(h_op, 'part', None), getattr(h_op, 't', None),
                                        getattr(gl_op, 't', None)), keep=(g, h_op, gl_op, weight_grads)) \
            if (acc and tgt_loop is not None and K.current_mode() != 'bf16s') else None
        if sd is not None:
            if isinstance(gl_op, K.F32Op):
                gl_op.bound()                                          # (measured on this stream: before the fork below)
            if isinstance(h_op, K.F32Op):
                h_op.bound()
            sd.side.wait_stream(sd.main)
            with sd():
                weight_grads()
            d_loop = None
        else:
            d_loop = weight_grads()
        return dh, None if acc else d_w, d_loop, None, None, None, None, None, None


class TableRows(object):
    """Deferred `table[idx]` (utils.py:239: h0 = ent_embeds[id]): handed to the first RGCN layer as g.ndata['h'], which
    then reads the entity table through idx instead of a materialised [N, D] copy (RGCNTableLayerFn)."""
    __slots__ = ('table', 'idx', 'plan')

    def __init__(self, table, idx, plan):
        self.table, self.idx, self.plan = table, idx, plan

    def materialise(self):
        return GatherRowsFn.apply(self.table, self.idx, self.plan)


class RGCNTableLayerFn(Function):
    """The FIRST RGCNBlockLayer of a pass (RGCN.py:33-51,79-94 on h0 = ent_embeds[id], utils.py:239) without ever
    materialising h0: h0 @ W_loop == (ent_embeds @ W_loop)[id], so the self-loop GEMM runs on the N_ent rows of the
    entity table (23 k at ICEWS18 sizes) instead of the N rows of the batch graph (92 k on the merged batch), and the
    gather-SpMM reads its source rows and its addend through g.node_ent from the two [N_ent, D] tables (18 MB each:
    cache resident) -- renet_rgcn_gather_items_table.  Backward: the transposed gather on the [N, D] gradient as in
    RGCNLayerFn, then ONE reduction to entity rows: d_ent += segsum(dh) + segsum(g_loop) @ W_loop^T and
    dW_loop = ent^T @ segsum(g_loop) -- again GEMMs over N_ent rows."""

    @staticmethod
    @_fwd_mode
    def forward(ctx, table, weight, loop_weight, g, reverse, relu, drop_p, seed):
        ctx.src_tab, ctx.src_w, ctx.src_loop = table, weight, loop_weight
        table, weight, loop_weight = _c(table), _c(weight), _c(loop_weight)
        shift = g.num_types // 2 if reverse else 0
        ew = K.gemm(table, loop_weight)                               # RGCN.py:35 on the entity table
        out = torch.empty(g.N, table.shape[1], device=table.device, dtype=torch.float32)
        K.rgcn_gather_items_table(table, g, weight, shift, ew, drop_p, seed, relu, out,
                                  table16=K.gather_weight_bf16(table), w16=K.gather_weight_bf16(weight))
        ctx.g, ctx.relu, ctx.drop_p, ctx.seed, ctx.shift = g, relu, drop_p, seed, shift
        ctx.save_for_backward(table, weight, loop_weight, out)
        return out

    @staticmethod
    @_bwd_mode
    def backward(ctx, g_out):
        table, weight, loop_weight, out = ctx.saved_tensors
        g = ctx.g
        tgt_tab, tgt_loop, tgt_w = grad_target(ctx.src_tab), grad_target(ctx.src_loop), grad_target(ctx.src_w)
        g_out = _c(g_out)
        n, d = out.shape
        dev = out.device
        gn = torch.empty(n, d, device=dev, dtype=torch.float32)
        g_loop = torch.empty(n, d, device=dev, dtype=torch.float32)
        K.rgcn_bwd_prep(g_out, out, g.norm, ctx.relu, ctx.drop_p, ctx.seed, gn, g_loop)
        dh = torch.empty(n, d, device=dev, dtype=torch.float32)
        pair_shift = (ctx.shift + g.num_types // 2) % g.num_types
        K.rgcn_gather_items(gn, g, weight, pair_shift, True, None, 0.0, 0, False, dh, use_norm=False,
                            w16=K.gather_weight_bf16(weight))
        acc = tgt_w is not None
        d_w = tgt_w if acc else torch.empty_like(weight)
        e_src_t = g.table_items()[3]
        sd = _deferred_scope(dev, (gn, e_src_t), keep=(g, table)) if acc else None      # (see RGCNLayerFn.backward)
        with (sd() if sd is not None else contextlib.nullcontext()):
            K.rgcn_bwd_w(table, gn, e_src_t, g.e_dst, g.chunk_ptr, g.chunk_type, g.n_chunks, g.type_chunk_ptr,
                         g.num_types, ctx.shift, d_w, beta=1.0 if acc else 0.0)
        # per-entity sums of the self-loop gradient, then the two self-loop GEMMs on N_ent rows
        gs = torch.zeros(table.shape, device=dev, dtype=torch.float32)
        d_tab = tgt_tab if tgt_tab is not None else torch.zeros(table.shape, device=dev, dtype=torch.float32)
        K.segment_add2(dh, g_loop, g.plan_node_ent, d_tab, gs)
        gs_op = K.operand(gs)
        K.gemm(gs_op, loop_weight, tb=True, out=d_tab, beta=1.0)          # += segsum(g_loop) @ W_loop^T
        if tgt_loop is not None:
            K.gemm(table, gs_op, ta=True, out=tgt_loop, beta=1.0)
            d_loop = None
        else:
            d_loop = K.gemm(table, gs_op, ta=True)
        return (None if tgt_tab is not None else d_tab), None if acc else d_w, d_loop, None, None, None, None, None


class SeqAssembleFn(Function):
    """Aggregator.py:139-165: packed GRU inputs X [S,4D], Xr [S,3D] with fused dropout."""

    @staticmethod
    @_fwd_mode
    def forward(ctx, h2, ent, rel, glob, g, drop_p, seed_x, seed_xr, _lazy_bf16=False):
        ctx.src_ent, ctx.src_rel = ent, rel
        h2, ent, rel, glob = _c(h2), _c(ent), _c(rel), _c(glob)
        if _lazy_bf16 and K.current_mode() == 'bf16s':
            # bf16-storage mode, internal callers only (RENet.loss_prepared*): X / Xr exist ONLY as bf16 operand
            # matrices; the fp32 tensors returned to autograd are uninitialised shells that carry them (K.operand
            # picks the attribute up) -- their values must never be read
            mx, mxr = K.seq_assemble_fwd_bf16(h2, ent, rel, glob, g.subj_row, g.row_ent, g.row_rel, g.glob_row,
                                              drop_p, seed_x, seed_xr)
            x, xr = K.lazy_shell(mx, h2.device), K.lazy_shell(mxr, h2.device)
        else:
            x, xr = K.seq_assemble_fwd(h2, ent, rel, glob, g.subj_row, g.row_ent, g.row_rel, g.glob_row,
                                       drop_p, seed_x, seed_xr)
        ctx.g, ctx.drop_p, ctx.seeds = g, drop_p, (seed_x, seed_xr)
        ctx.shapes = (h2.shape, ent.shape, rel.shape)
        return x, xr

    @staticmethod
    @_bwd_mode
    def backward(ctx, dx, dxr):
        g = ctx.g
        d = ctx.shapes[0][1]
        tgt_ent, tgt_rel = grad_target(ctx.src_ent), grad_target(ctx.src_rel)
        d_rows, d_ent_seq, d_rel_seq = K.seq_assemble_bwd(_c(dx), _c(dxr), g.step_off, g.L, g.B, d, ctx.drop_p,
                                                          *ctx.seeds)
        dev = dx.device
        d_h2 = torch.zeros(ctx.shapes[0], device=dev, dtype=torch.float32)
        K.segment_add(d_rows, g.plan_subj_row, d_h2)
        d_ent = d_rel = None
        if tgt_ent is not None:
            K.segment_add(d_ent_seq, g.plan_s, tgt_ent)           # per-sequence sums, keyed by s[perm]
        else:
            d_ent = torch.zeros(ctx.shapes[1], device=dev, dtype=torch.float32)
            K.segment_add(d_ent_seq, g.plan_s, d_ent)
        if tgt_rel is not None:
            K.segment_add(d_rel_seq, g.plan_r, tgt_rel)
        else:
            d_rel = torch.zeros(ctx.shapes[2], device=dev, dtype=torch.float32)
            K.segment_add(d_rel_seq, g.plan_r, d_rel)
        return d_h2, d_ent, d_rel, None, None, None, None, None, None


class GRUFn(Function):
    """nn.GRU (1 layer, h0 = 0) on a packed sequence, returning h_n zero-padded to `total_rows`
    (model.py:86-88).  x: [S, I] packed time-major; step_off: ctypes int32[L+1] on the host."""

    @staticmethod
    @_fwd_mode
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, step_off, total_rows):
        x, w_ih, w_hh, b_ih, b_hh = _c(x), _c(w_ih), _c(w_hh), _c(b_ih), _c(b_hh)
        hdim = w_hh.shape[1]
        gi = K.gemm(x, w_ih, tb=True, bias=b_ih)                         # [S, 3H]
        full, saved = K.gru_fwd(gi, step_off, hdim, w_hh, b_hh, out_rows=total_rows)     # zero rows past nnz
        nnz = int(step_off[1] - step_off[0]) if len(step_off) > 1 else 0
        ctx.step_off, ctx.nnz, ctx.hdim = step_off, nnz, hdim
        ctx.save_for_backward(x, w_ih, w_hh, saved)
        return full.unsqueeze(0)                 # h_n layout of nn.GRU: [1, B, H]

    @staticmethod
    @_bwd_mode
    def backward(ctx, dh):
        x, w_ih, w_hh, saved = ctx.saved_tensors
        hdim, nnz = ctx.hdim, ctx.nnz
        dh_last = _c(dh[0, :nnz])
        d_gi, d_gh = K.gru_bwd(dh_last, ctx.step_off, hdim, w_hh, saved)
        s = x.shape[0]
        dx = K.gemm(d_gi, w_ih)                                          # [S, I]
        d_wih = K.gemm(d_gi, x, ta=True)
        d_bih = K.colsum(d_gi)
        h_prev = saved[:, 4 * hdim:]
        d_whh = K.gemm(d_gh, h_prev, ta=True)
        d_bhh = K.colsum(d_gh)
        return dx, d_wih, d_whh, d_bih, d_bhh, None, None


class MultiGRUFn(Function):
    """n <= 4 of RE-Net's history encoders in ONE persistent launch per direction of time (model.py:86,94):
    `encoder` (4D->D) and `encoder_r` (3D->D) of one pass share a packed layout; the subject and the object pass of
    a training step are independent until their losses are added, so a step may run all four side by side
    (RENet.loss_prepared_pair).  Input projections are one GEMM per problem.
    apply(step_offs, total_rows, live_cols, x_0, w_ih_0, w_hh_0, b_ih_0, b_hh_0, x_1, ...) -> (h_n_0 [1, rows_0, H], ...)
    live_cols: None, or per problem the number of leading columns of dX the caller reads (None = all)."""

    @staticmethod
    @_fwd_mode
    def forward(ctx, step_offs, total_rows, live_cols, *ts):
        n = len(ts) // 5
        ctx.live_cols = live_cols
        ctx.src_w = [(ts[5 * k + 1], ts[5 * k + 2]) for k in range(n)]
        ctx.src_b = [(ts[5 * k + 3], ts[5 * k + 4]) for k in range(n)]
        ts = [_c(t) for t in ts]
        xs, w_ihs, w_hhs = ts[0::5], ts[1::5], ts[2::5]
        b_ihs, b_hhs = ts[3::5], ts[4::5]
        hdim = w_hhs[0].shape[1]
        x_ops = [K.operand(x) for x in xs]                              # (bf16 mode: packed once, reused by dW_ih)
        # the input projections of the problems are independent: odd problems on the side stream (_Side)
        sd = _Side(xs[0].device)
        gis = [None] * n
        with sd():
            for k in range(1, n, 2):
                gis[k] = K.gemm(x_ops[k], w_ihs[k], tb=True, bias=b_ihs[k])
        for k in range(0, n, 2):
            gis[k] = K.gemm(x_ops[k], w_ihs[k], tb=True, bias=b_ihs[k])
        sd.join()
        ctx.x_ops = x_ops if any(K.is_handle(x) for x in x_ops) else None
        hs, svs = K.gru_fwd_layouts(gis, step_offs, hdim, w_hhs, b_hhs, total_rows)     # rows past nnz are zero
        ctx.step_offs, ctx.hdim, ctx.n = step_offs, hdim, n
        ctx.nnz = [int(o[1] - o[0]) if len(o) > 1 else 0 for o in step_offs]
        ctx.save_for_backward(*(list(xs) + list(w_ihs) + list(w_hhs) + list(svs)))
        if debug_tap is not None:
            for k, h in enumerate(hs):
                debug_tap('h_n' if k % 2 == 0 else 'q_n', h)
        return tuple(h.unsqueeze(0) for h in hs)

    @staticmethod
    @_bwd_mode
    def backward(ctx, *dhs):
        n, hdim = ctx.n, ctx.hdim
        sv_ = ctx.saved_tensors
        xs, w_ihs, w_hhs, svs = sv_[:n], sv_[n:2 * n], sv_[2 * n:3 * n], sv_[3 * n:4 * n]
        d_gis, d_ghs = K.gru_bwd_layouts([_c(dh[0, :nz]) for dh, nz in zip(dhs, ctx.nnz)], ctx.step_offs, hdim,
                                         list(w_hhs), list(svs), out_bf16=(K.current_mode() == 'bf16s'))
        def param_grads(k, dgi_op):
            """dW_ih, dW_hh, db_ih, db_hh of problem k: consumed by the optimizer only."""
            t_ih, t_hh = (grad_target(t) for t in ctx.src_w[k])
            t_bi, t_bh = (grad_target(t) for t in ctx.src_b[k])
            dgi, dgh, xx, s_ = d_gis[k], d_ghs[k], xs[k], svs[k]
            x_op = ctx.x_ops[k] if ctx.x_ops is not None else xx
            dwi = dwh = dbi = dbh = None
            if t_ih is not None:
                K.gemm(dgi_op, x_op, ta=True, out=t_ih, beta=1.0)
            else:
                dwi = K.gemm(dgi_op, x_op, ta=True)
            # |dGh| <= |dGi| elementwise (equal but for the n gate's factor r in (0, 1)) and |h| <= 1: both operand
            # bounds of the f16x3 GEMM are known without a pass over the tensors
            dgh_op = K.operand_like(dgh, dgi_op)
            hp_op = K.operand(s_[:, 4 * hdim:], bound=K.const_bound(1.0, s_.device))
            if t_hh is not None:
                K.gemm(dgh_op, hp_op, ta=True, out=t_hh, beta=1.0)
            else:
                dwh = K.gemm(dgh_op, hp_op, ta=True)
            if t_bi is not None:
                K.colsum(dgi, out=t_bi, beta=1.0)
            else:
                dbi = K.colsum(dgi)
            if t_bh is not None:
                K.colsum(dgh, out=t_bh, beta=1.0)
            else:
                dbh = K.colsum(dgh)
            return [dwi, dwh, dbi, dbh]

        def input_grad(k, dgi_op):
            xx = xs[k]
            live = ctx.live_cols[k] if ctx.live_cols is not None else None
            if live is not None and live < xx.shape[1]:
                # the caller never reads dX beyond column `live` (RE-Net: the last D input columns are the global
                # embedding, a constant, Aggregator.py:150-155): contract only the live columns of W_ih; the rest
                # of dX stays unwritten
                dxx = torch.empty_like(xx)
                K.gemm(dgi_op, w_ihs[k][:, :live], out=dxx[:, :live])
                dxx[:, live:].zero_()         # defined values for any other consumer (hooks, detect_anomaly): 6 MB
                return dxx
            return K.gemm(dgi_op, w_ihs[k])

        def problem(k):
            dgi_op = K.operand(d_gis[k])                               # consumed by dW_ih and dX
            return [input_grad(k, dgi_op)] + param_grads(k, dgi_op)

        all_in_place = all(grad_target(t) is not None for k in range(n) for t in ctx.src_w[k] + ctx.src_b[k])
        sd0 = _Side(d_gis[0].device if not isinstance(d_gis[0], K.BF16Mat) else d_gis[0].p.device)
        if DEFER_WEIGHT_GRADS and sd0.on and all_in_place and K.current_mode() != 'bf16s':
            # every parameter gradient accumulates in place and somebody joins before the optimizer reads it: dX of all
            # problems on this stream (what the rest of the backward pass waits for), all parameter gradients on the side
            # stream, NOT joined here
            ops_ = [K.operand(d_gis[k]) for k in range(n)]
            if K.current_mode() == 'f16x3':
                for o in ops_:
                    o.bound()                                          # measured on THIS stream, before the fork
            sd = _Side(sd0.main.device)                                # (fork: the side stream sees d_gis / bounds)
            res = [None] * n
            with sd():
                for k in range(n):
                    param_grads(k, ops_[k])
                _gru_grads_done(ctx)                                   # (on the side stream: the reducer orders behind it)
            live_ops = ops_ + (list(ctx.x_ops) if ctx.x_ops is not None else [])
            for t in list(d_gis) + list(d_ghs) + list(xs) + list(svs) + \
                    [getattr(o, 'part', None) for o in live_ops] + [getattr(o, 't', None) for o in live_ops]:
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(sd.side)
            for k in range(n):
                res[k] = [input_grad(k, ops_[k]), None, None, None, None]
            _deferred.append((sd.main, sd.side, (list(d_gis), list(d_ghs), list(xs), list(svs), live_ops)))
            out = [None, None, None]
            for k in range(n):
                out += res[k]
            return tuple(out)

        # problems with DIFFERENT parameters are independent of each other (encoder / encoder_r; two problems of the
        # same encoder accumulate into the same gradient buffers and stay on one stream): the second parameter set's
        # problems go to the side stream
        keys = [id(ctx.src_w[k][0]) for k in range(n)]
        uniq = list(dict.fromkeys(keys))
        on_side = [len(uniq) > 1 and keys[k] == uniq[1] for k in range(n)]
        res = [None] * n
        sd = sd0
        with sd():
            for k in range(n):
                if on_side[k]:
                    res[k] = problem(k)
        for k in range(n):
            if not on_side[k]:
                res[k] = problem(k)
        sd.join()
        if all_in_place:
            _gru_grads_done(ctx)
        out = [None, None, None]
        for k in range(n):
            out += res[k]
        return tuple(out)


def _gru_grads_done(ctx):
    """MultiGRUFn.backward accumulated into every encoder parameter's .grad: tell whoever registered (the reducer's middle
    bucket).  A parameter shared by two problems of one call (the subject / object pair) is reported once per problem."""
    if not _grad_done_hooks:
        return
    for k in range(len(ctx.src_w)):
        for t in list(ctx.src_w[k]) + list(ctx.src_b[k]):
            grad_done(t)


def dual_gru(x, xr, enc, enc_r, step_off, total_rows):
    """Both encoders of ONE pass: -> (h_n [1, rows, H], q_n [1, rows, H]).  x = [h2 | ent | rel | glob] and
    xr = [h2 | ent | glob] come from SeqAssembleFn, whose backward reads every column but the trailing glob block."""
    h = enc.weight_hh_l0.shape[1]
    return MultiGRUFn.apply([step_off, step_off], [total_rows, total_rows], [x.shape[1] - h, xr.shape[1] - h],
                            x, enc.weight_ih_l0, enc.weight_hh_l0, enc.bias_ih_l0, enc.bias_hh_l0,
                            xr, enc_r.weight_ih_l0, enc_r.weight_hh_l0, enc_r.bias_ih_l0, enc_r.bias_hh_l0)


def _head_forward(a, ia, hmid, c, ic, weight, bias, target, drop_p, seed, grad_scale, need_grad, row_loss=None):
    """One score head up to the per-row losses: [a[ia] | hmid | c[ic]] -> dropout -> Linear -> CE.
    -> (row_loss[B], feat, logits-or-gradient buffer, bf16 / planes gradient matrix or None, operand handle of feat or None)"""
    feat = K.concat3_fwd(a, ia, hmid, c, ic, drop_p, seed)
    if need_grad and debug_tap is None and K.use_planes(weight, weight.shape[0]):
        # PLANES path (round 6, the entity head): every operand of the head's three GEMMs is split into its bf16 terms ONCE
        # -- the weight per optimizer step, feat here, the CE gradient by the kernel that computes it -- and the GEMMs run
        # without conversion work (csrc/gemm_p6.h).  The fp32 logits live in a scratch buffer with 16-byte aligned rows
        # and are dropped after the loss; no fp32 copy of the gradient exists.
        b_, c_ = feat.shape[0], weight.shape[0]
        w_pl = K.weight_planes(weight)
        logits = torch.empty(b_, (c_ + 3) & ~3, device=feat.device, dtype=torch.float32)[:, :c_]
        # feat is split ONCE, with a ones column: dW's GEMM then yields the bias gradient as one more output column
        # (col_out); the logits GEMM reads the same planes with K = 3D (the ones meet the zero k padding of the weight's
        # planes and add exactly 0)
        feat_pl1 = K.pack_planes(feat, ones_col=True)
        K.gemm_planes(K.PlanesMat(feat_pl1.p, feat_pl1.R, feat_pl1.C - 1), w_pl, tb=True, bias=bias, out=logits)
        row_loss, dl_pl = K.softmax_ce_planes(logits, target, grad_scale, row_loss=row_loss)
        return row_loss, feat, feat.new_empty(0), dl_pl, feat_pl1
    feat_op = K.operand(feat)                                        # (bf16 mode: packed once, reused by dW)
    logits = K.gemm(feat_op, weight, tb=True, bias=bias)             # [B, C]
    if debug_tap is not None:
        debug_tap('logits', logits)
    dl_bf16 = None
    if need_grad and K.current_mode() == 'bf16s':
        # bf16-storage mode: the gradient is written as a bf16 operand matrix, the fp32 logits are dropped
        row_loss, dl_bf16 = K.softmax_ce_bf16(logits, target, grad_scale, row_loss=row_loss)
        logits = feat.new_empty(0)
    else:
        row_loss = K.softmax_ce(logits, target, grad_scale, need_grad, row_loss=row_loss)
    return row_loss, feat, logits, dl_bf16, (feat_op if K.is_handle(feat_op) else None)


def _scale_ce_gradient(dlogits, g, grad_scale):
    """dlogits *= g (the upstream scalar, device memory) -> the bound |g| * grad_scale on max |dlogits| as a 1-element
    device tensor in f16x3 mode (those GEMMs scale their operands by a bound on the tensor's magnitude; |softmax -
    onehot| <= 1, so this one is known without a pass over the 188 MB), else None."""
    if isinstance(dlogits, K.PlanesMat):
        return None                      # (the planes GEMMs take the upstream scalar as alpha_dev: no pass over the gradient)
    if K.current_mode() == 'f16x3' and not isinstance(dlogits, K.BF16Mat) and dlogits.is_cuda:
        return K.scale_by_device_scalar(dlogits, g, bound_in=float(grad_scale))
    K.scale_by_device_scalar(dlogits, g)
    return None


def _head_backward(dlogits, feat, feat_op, weight, t_w, t_b, bias_side=None, bound=None, g=None):
    """The three gradient products of one score head from its (already scaled) CE gradient -> (dfeat, d_w, d_b).
    bias_side: a _Side whose stream takes the bias column sum (bandwidth bound, next to the matrix-bound GEMMs that
    read the same gradient).  bound: see _scale_ce_gradient.  g: the upstream scalar (device), applied here when the
    gradient arrives as planes (it was NOT scaled then)."""
    if isinstance(dlogits, K.PlanesMat):
        w_pl = K.weight_planes(weight)
        dfeat = K.gemm_planes(dlogits, w_pl, alpha_dev=g)            # [B, parts*D]: contraction over the classes
        # dW and db in ONE product: [C, parts*D | 1] = dl^T @ [feat | 1]
        acc = t_w is not None and t_b is not None
        d_w = t_w if acc else torch.empty_like(weight)
        d_b = t_b if acc else torch.empty(weight.shape[0], device=weight.device, dtype=torch.float32)
        K.gemm_planes(dlogits, feat_op, ta=True, out=d_w, col_out=d_b, alpha_dev=g, beta=1.0 if acc else 0.0)
        return dfeat, (None if acc else d_w), (None if acc else d_b)
    dl_op = K.operand(dlogits, bound=bound)                          # consumed by dfeat and dW
    f_op = feat_op if feat_op is not None else feat

    def bias_grad():
        if t_b is not None:
            K.colsum(dlogits, out=t_b, beta=1.0)
            return None
        return K.colsum(dlogits)
    if bias_side is not None:
        with bias_side():
            d_b = bias_grad()
    dfeat = K.gemm(dl_op, weight)                                    # [B, parts*D]
    # (the weight gradient stays BEHIND dfeat on this stream: next to it on the side stream the two chip-filling GEMMs
    # were measured 2 % slower per step than back to back, tools/sessions/r04_s11.sh)
    if t_w is not None:
        K.gemm(dl_op, f_op, ta=True, out=t_w, beta=1.0)
        d_w = None
    else:
        d_w = K.gemm(dl_op, f_op, ta=True)
    if bias_side is None:
        d_b = bias_grad()
    return dfeat, d_w, d_b


class HeadCEFn(Function):
    """model.py:89-91 / 98-100: mean CE( Linear( dropout([a[ia] | hmid | c[ic]]) ), target ).
    The logits never make a second HBM round trip as a separate softmax: the CE kernel turns them
    into (softmax - onehot)/B in place, which the backward GEMMs then consume."""

    @staticmethod
    @_fwd_mode
    def forward(ctx, a, ia, hmid, c, ic, weight, bias, target, plan_a, plan_c, drop_p, seed, loss_scale=1.0):
        ctx.srcs = (a, c, weight, bias)
        a, hmid, weight, bias = _c(a), _c(hmid), _c(weight), _c(bias)
        c = _c(c) if c is not None else None
        b, d = hmid.shape
        need_grad = any(ctx.needs_input_grad)
        # loss_scale (2 for the merged batch of both passes: sum of two B-row means = 2 x the 2B-row mean) goes into
        # the gradient the CE kernel writes, so that the upstream scalar stays 1 and the 188 MB are not rescaled
        ctx.grad_scale = float(loss_scale) / b
        row_loss, feat, logits, ctx.dl_bf16, feat_op = _head_forward(a, ia, hmid, c, ic, weight, bias, target, drop_p,
                                                                     seed, ctx.grad_scale, need_grad)
        ctx.meta = (d, 3 if c is not None else 2, drop_p, seed, plan_a, plan_c, a.shape,
                    c.shape if c is not None else None)
        if need_grad:
            ctx.save_for_backward(feat, logits, weight)
            ctx.consumed = False
            ctx.feat_op = feat_op
        return row_loss.mean() if loss_scale == 1.0 else row_loss.mean() * float(loss_scale)

    @staticmethod
    @_bwd_mode
    def backward(ctx, g):
        feat, dlogits, weight = ctx.saved_tensors
        d, parts, drop_p, seed, plan_a, plan_c, a_shape, c_shape = ctx.meta
        t_a, t_c, t_w, t_b = [grad_target(t) for t in ctx.srcs]
        if ctx.consumed:
            raise RuntimeError('HeadCEFn: the saved (softmax - onehot) buffer was scaled in place by the first '
                               'backward pass; a second pass over the same graph is not supported')
        ctx.consumed = True
        # every gradient below is linear in dlogits: fold the upstream scalar (1 for `loss_s + loss_o`, 0.1 for
        # the relation head) into it ONCE, from device memory, instead of scaling three results
        if ctx.dl_bf16 is not None:
            dlogits = ctx.dl_bf16
        bound = _scale_ce_gradient(dlogits, g, ctx.grad_scale)
        sd = _Side(feat.device)
        dfeat, d_w, d_b = _head_backward(dlogits, feat, ctx.feat_op, weight, t_w, t_b, bias_side=sd, bound=bound, g=g)
        sd.join()
        if t_w is not None and t_b is not None:
            grad_done(ctx.srcs[2])
            grad_done(ctx.srcs[3])
        da_rows, dh, dc_rows = K.concat3_bwd(dfeat, d, parts, drop_p, seed)
        d_a = d_c = None
        if t_a is not None:
            K.segment_add(da_rows, plan_a, t_a)
        else:
            d_a = torch.zeros(a_shape, device=g.device, dtype=torch.float32)
            K.segment_add(da_rows, plan_a, d_a)
        if parts == 3:
            if t_c is not None:
                K.segment_add(dc_rows, plan_c, t_c)
            else:
                d_c = torch.zeros(c_shape, device=g.device, dtype=torch.float32)
                K.segment_add(dc_rows, plan_c, d_c)
        return d_a, None, dh, d_c, None, d_w, d_b, None, None, None, None, None, None


_loss_weights = {}


def _loss_weight_vector(b, w1, w2, device):
    key = (b, float(w1), float(w2), str(device))
    v = _loss_weights.get(key)
    if v is None:
        if len(_loss_weights) > 16:
            _loss_weights.clear()
        v = torch.cat((torch.full((b,), float(w1)), torch.full((b,), float(w2)))).to(device)
        _loss_weights[key] = v
    return v


class DualHeadCEFn(Function):
    """Both score heads of one pass and their weighted sum (model.py:89-103) as ONE Function:
        loss_scale * ( mean CE(Linear([ent[s] | s_h | rel[r]]), o)  +  rel_weight * mean CE(Linear_r([ent[s] | s_q]), r) )
    The relation head (C = 2R classes: three GEMMs of 16-32 workgroups and a handful of few-microsecond kernels per
    direction of the pass) runs on the side stream next to the entity head's chip-filling GEMMs, in forward and in
    backward; so does the entity head's bias column sum.  The two heads gather the SAME rows ent[s]: their row
    gradients are added before ONE segmented scatter-add.  Values: those of the two HeadCEFn calls it replaces (the
    weight rel_weight is folded into the relation head's CE gradient instead of being applied by autograd)."""

    @staticmethod
    @_fwd_mode
    def forward(ctx, a, ia, h1, c, ic, w1, b1, target1, h2, w2, b2, target2, plan_a, plan_c, drop_p, seed1, seed2,
                loss_scale=1.0, rel_weight=0.1, row_tap=None):
        """row_tap: a list that receives (row losses [2b]: entity head rows then relation head rows, s1, s2) -- the weights
        with which the returned scalar sums them (RENet.forward's fused directions split the merged loss by direction)."""
        ctx.srcs = (a, c, w1, b1, w2, b2)
        a, h1, h2, c, w1, b1, w2, b2 = (_c(t) for t in (a, h1, h2, c, w1, b1, w2, b2))
        b, d = h1.shape
        need_grad = any(ctx.needs_input_grad)
        s1, s2 = float(loss_scale) / b, float(loss_scale) * float(rel_weight) / b
        rl = torch.empty(2 * b, device=h1.device, dtype=torch.float32)
        def head2():
            return _head_forward(a, ia, h2, None, None, w2, b2, target2, drop_p, seed2, s2, need_grad, row_loss=rl[b:])
        sd = _Side(h1.device)
        if debug_tap is None:
            with sd():
                _, feat2, lg2, dl2, fop2 = head2()
        _, feat1, lg1, dl1, fop1 = _head_forward(a, ia, h1, c, ic, w1, b1, target1, drop_p, seed1, s1, need_grad,
                                                 row_loss=rl[:b])
        if debug_tap is not None:                            # test hook: the entity head reports first
            _, feat2, lg2, dl2, fop2 = head2()
        sd.join()
        ctx.meta = (d, drop_p, seed1, seed2, plan_a, plan_c, a.shape, c.shape)
        ctx.grad_scales = (s1, s2)
        if need_grad:
            ctx.save_for_backward(feat1, lg1, w1, feat2, lg2, w2)
            ctx.consumed = False
            ctx.bf16 = (dl1, fop1, dl2, fop2)
        if row_tap is not None:
            row_tap.append((rl.detach(), s1, s2))
        return torch.dot(rl, _loss_weight_vector(b, s1, s2, h1.device))

    @staticmethod
    @_bwd_mode
    def backward(ctx, g):
        feat1, dl1, w1, feat2, dl2, w2 = ctx.saved_tensors
        d, drop_p, seed1, seed2, plan_a, plan_c, a_shape, c_shape = ctx.meta
        t_a, t_c, t_w1, t_b1, t_w2, t_b2 = [grad_target(t) for t in ctx.srcs]
        if ctx.consumed:
            raise RuntimeError('DualHeadCEFn: the saved (softmax - onehot) buffers were scaled in place by the first '
                               'backward pass; a second pass over the same graph is not supported')
        ctx.consumed = True
        b1_, fop1, b2_, fop2 = ctx.bf16
        dl1 = b1_ if b1_ is not None else dl1
        dl2 = b2_ if b2_ is not None else dl2
        bound1 = _scale_ce_gradient(dl1, g, ctx.grad_scales[0])      # upstream scalar (1 in RE-Net's training step)
        sd = _Side(feat1.device)
        # (deferring the entity head's dW / db to the side stream until opt.step(), like the GRU parameter gradients, was
        # measured: no gain -- 2.975 vs 2.987 ms per step, tools/sessions/r04_s18.sh: a chip-filling GEMM only competes
        # with the GRU / RGCN backward kernels it would run beside)
        with sd():
            bound2 = _scale_ce_gradient(dl2, g, ctx.grad_scales[1])
            dfeat2, d_w2, d_b2 = _head_backward(dl2, feat2, fop2, w2, t_w2, t_b2, bound=bound2, g=g)
            da2, dh2, _ = K.concat3_bwd(dfeat2, d, 2, drop_p, seed2)
        dfeat1, d_w1, d_b1 = _head_backward(dl1, feat1, fop1, w1, t_w1, t_b1, bias_side=sd, bound=bound1, g=g)
        da1, dh1, dc1 = K.concat3_bwd(dfeat1, d, 3, drop_p, seed1)
        sd.join()
        if t_w1 is not None and t_b1 is not None:
            grad_done(ctx.srcs[2])
            grad_done(ctx.srcs[3])
        if t_w2 is not None and t_b2 is not None:
            grad_done(ctx.srcs[4])
            grad_done(ctx.srcs[5])
        da1.add_(da2)                                        # same rows ent[s] in both heads: one scatter-add
        d_a = d_c = None
        if t_a is not None:
            K.segment_add(da1, plan_a, t_a)
        else:
            d_a = torch.zeros(a_shape, device=g.device, dtype=torch.float32)
            K.segment_add(da1, plan_a, d_a)
        if t_c is not None:
            K.segment_add(dc1, plan_c, t_c)
        else:
            d_c = torch.zeros(c_shape, device=g.device, dtype=torch.float32)
            K.segment_add(dc1, plan_c, d_c)
        return (d_a, None, dh1, d_c, None, d_w1, d_b1, None, dh2, d_w2, d_b2, None) + (None,) * 8


class SegmentPoolFn(Function):
    """dgl.max_nodes / mean_nodes over the member graphs (Aggregator.py:58-61)."""

    @staticmethod
    @_fwd_mode
    def forward(ctx, h, seg_ptr, num_graphs, is_max):
        h = _c(h)
        out, arg = K.segment_pool_fwd(h, seg_ptr, num_graphs, is_max)
        ctx.meta = (seg_ptr, arg, num_graphs, is_max, h.shape[0])
        return out

    @staticmethod
    @_bwd_mode
    def backward(ctx, dout):
        seg_ptr, arg, num_graphs, is_max, n = ctx.meta
        return K.segment_pool_bwd(_c(dout), seg_ptr, arg, num_graphs, is_max, n), None, None, None


class DropoutFn(Function):
    """Counter-based inverted dropout (Aggregator.py:69); the mask is regenerated in backward."""

    @staticmethod
    @_fwd_mode
    def forward(ctx, x, drop_p, seed):
        ctx.meta = (drop_p, seed)
        return K.dropout(_c(x), drop_p, seed)

    @staticmethod
    @_bwd_mode
    def backward(ctx, g):
        return K.dropout(_c(g), *ctx.meta), None, None


class LinearFn(Function):
    """y = x @ W^T + b on the MFMA GEMM (nn.Linear forward/backward; global_model.py:52)."""

    @staticmethod
    @_fwd_mode
    def forward(ctx, x, weight, bias):
        x, weight = _c(x), _c(weight)
        ctx.save_for_backward(x, weight)
        return K.gemm(x, weight, tb=True, bias=_c(bias))

    @staticmethod
    @_bwd_mode
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = _c(g)
        return (K.gemm(g, weight), K.gemm(g, x, ta=True), K.colsum(g))


def host_offsets(step_off):
    """int array -> ctypes int32 array kept alive by the caller (the GRU entry points read it on the host)."""
    arr = np.ascontiguousarray(step_off, dtype=np.int32)
    return (ctypes.c_int32 * len(arr))(*arr.tolist())
