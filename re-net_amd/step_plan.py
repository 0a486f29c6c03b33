"""The merged training step as ONE launch list issued from C (csrc/step.cpp, include/renet_hip.h renet_step_*).

RENet.loss_prepared_both normally runs through five autograd Functions (ops.py) that make ~55 C-ABI calls per step from
Python -- 2 ms of launching-thread time for a step the MI355X runs in 2.7 ms.  Here the same launch sequence is two C calls:
`StepFn.forward` fills three plain structs (the model's parameter / gradient pointers, the batch's device arrays -- cached on
the prepared batch -- and the run's seeds / workspace / streams) and calls renet_step_forward; `StepFn.backward` calls
renet_step_backward with the upstream scalar.  Gradients accumulate straight into the existing .grad buffers (as the autograd
path does with ops.INPLACE_GRADS), activations live in one workspace tensor that the Function owns between the two calls.
Every kernel receives exactly the arguments of the autograd path: losses and gradients are bit-identical
(tests/test_gpu_step_plan.py).  Mirrors one iteration of the reference's train.py:136-139.

Used when: default fp32-class mode (bf16x6), no kernel timer / debug tap, RENET_STEP_PLAN != 0.  A parameter that owns a
contiguous .grad (parallel.FlatGrads / HipAdam, zero_grad(set_to_none=False)) is accumulated into in place; for parameters
without one (train.py's loop: torch's zero_grad() drops them) the gradients are materialised in one zeroed flat buffer and
returned to autograd.  Anything else takes the autograd path."""
import ctypes
import os
import weakref

import torch
from torch.autograd import Function

import ops
import renet_hip as K

ENABLED = os.environ.get('RENET_STEP_PLAN', '1') != '0'

# per-model caches (the parameter tuple, the filled RenetStepModel) live HERE, keyed weakly by the module -- not in the
# module's __dict__: a ctypes struct there would make copy.deepcopy(model) / torch.save(model) fail
_cache = weakref.WeakKeyDictionary()


def _slot(net):
    c = _cache.get(net)
    if c is None:
        c = _cache[net] = {}
    return c

_P = ctypes.c_void_p
_I = ctypes.c_int


class SegPlanC(ctypes.Structure):
    _fields_ = [('order', _P), ('seg_ptr', _P), ('target', _P), ('num_segments', _I)]


_PARAMS = ('ent', 'rel', 'w1', 'loop1', 'w2', 'loop2', 'wih', 'whh', 'bih', 'bhh', 'wih_r', 'whh_r', 'bih_r', 'bhh_r', 'lin_w',
           'lin_b', 'linr_w', 'linr_b')


class StepModelC(ctypes.Structure):
    _fields_ = [('D', _I), ('num_ent', _I), ('T', _I), ('C2', _I), ('drop_p', ctypes.c_float)] + \
               [(n, _P) for n in _PARAMS] + [('g_' + n, _P) for n in _PARAMS] + \
               [('glob', _P), ('lin_w_planes', _P), ('lin_w_plane', ctypes.c_size_t), ('lin_w_ld', _I)]


_BATCH_INTS = ('N', 'E', 'nA', 'S', 'B', 'L', 'n_items', 'n_groups', 'n_groups_out', 'n_heavy', 'n_heavy_out', 'n_chunks',
               'n_chunks2')
_BATCH_PTRS = ('step_off_host', 'node_ent', 'row_ptr', 'col', 'etype', 'norm', 'it_src', 'it_type', 'grp_ptr', 'heavy_rows',
               'heavy_rows_out', 'it_src_t', 'it_type_t', 'col_t', 'e_src_t', 'e_src', 'e_dst', 'chunk_ptr', 'chunk_type',
               'type_chunk_ptr', 'e_src2', 'e_dst2', 'chunk_ptr2', 'chunk_type2', 'type_chunk_ptr2', 'subj_row', 'row_ent',
               'row_rel', 'glob_row', 'step_off', 's_idx', 'r_idx', 'ent_label', 'rel_label')


class StepBatchC(ctypes.Structure):
    _fields_ = [(n, _I) for n in _BATCH_INTS] + [(n, _P) for n in _BATCH_PTRS] + \
               [('plan_node_ent', SegPlanC), ('plan_subj_row', SegPlanC), ('plan_s', SegPlanC), ('plan_r', SegPlanC)]


class StepRunC(ctypes.Structure):
    _fields_ = [('seed_rgcn1', ctypes.c_uint64), ('seed_rgcn2', ctypes.c_uint64), ('seed_x', ctypes.c_uint64),
                ('seed_xr', ctypes.c_uint64), ('seed_head1', ctypes.c_uint64), ('seed_head2', ctypes.c_uint64),
                ('scale_ent', ctypes.c_float), ('scale_rel', ctypes.c_float), ('workspace', _P),
                ('workspace_bytes', ctypes.c_size_t), ('stream', _P), ('side_stream', _P)]


def model_params(net):
    """The 18 parameters of a RENet in the order of RenetStepModel (cached on the module: nn.Module attribute lookups are
    a dozen microseconds per step otherwise; re-derived when a Parameter object was replaced)."""
    c = _slot(net).get('params')
    if c is not None and c[0] is net._parameters.get('ent_embeds') and c[14] is net.linear._parameters.get('weight'):
        return c
    a, e, er = net.aggregator, net.encoder, net.encoder_r
    c = (net.ent_embeds, net.rel_embeds, a.rgcn1.weight, a.rgcn1.loop_weight, a.rgcn2.weight, a.rgcn2.loop_weight,
         e.weight_ih_l0, e.weight_hh_l0, e.bias_ih_l0, e.bias_hh_l0, er.weight_ih_l0, er.weight_hh_l0, er.bias_ih_l0,
         er.bias_hh_l0, net.linear.weight, net.linear.bias, net.linear_r.weight, net.linear_r.bias)
    _slot(net)['params'] = c
    _slot(net).pop('model', None)
    return c


def _grad_ok(p):
    g = p.grad
    return g is not None and g.is_cuda and g.dtype == torch.float32 and g.is_contiguous() and g.shape == p.shape


def _pointers(params):
    """(parameter pointers, gradient pointers) -- 0 for a missing gradient."""
    return tuple(p.data_ptr() for p in params), tuple(p.grad.data_ptr() if p.grad is not None else 0 for p in params)


def eligible(net, prep, row_tap=None):
    """Whether loss_prepared_both(prep) can run as the C launch list right now."""
    if not ENABLED or prep is None or K._timer is not None or ops.debug_tap is not None or not torch.is_grad_enabled():
        return False
    if K.current_mode() != 'bf16x6' or os.environ.get('RENET_GRU', '') == 'steps':
        return False
    g = prep.g
    if g is None or not hasattr(g, 'grp_ptr') or getattr(g, 'nA', None) is None:
        return False
    ps = model_params(net)
    ptrs = _pointers(ps)
    ent = _slot(net).get('model')
    if ent is None or ent[0] != ptrs:
        # first step, or a tensor was replaced: validate everything once for this set of pointers.  A parameter either owns
        # a usable .grad buffer (accumulated into in place) or none at all (train.py's loop: torch's zero_grad() sets it to
        # None -- its gradient is then materialised in backward and handed to autograd).
        if not all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.requires_grad and
                   (p.grad is None or _grad_ok(p)) for p in ps):
            return False
        _slot(net)['model'] = (ptrs, None)
    d = net.h_dim
    return max(g.N, net.ent_embeds.shape[0]) * d * 4 < (1 << 31) and g.S > 0 and g.L > 0


def _model_struct(net, params, planes):
    """RenetStepModel for the current pointers (the struct is kept while they do not change)."""
    ptrs = _pointers(params)
    ent = _slot(net).get('model')
    m = ent[1] if (ent is not None and ent[0] == ptrs) else None
    if m is None:
        m = StepModelC()
        m.D, m.num_ent, m.T, m.C2 = net.h_dim, net.ent_embeds.shape[0], net.rel_embeds.shape[0], net.linear_r.weight.shape[0]
        for n, pp, gp in zip(_PARAMS, ptrs[0], ptrs[1]):
            setattr(m, n, pp)
            setattr(m, 'g_' + n, gp or None)
        _slot(net)['model'] = (ptrs, m)
    m.drop_p = float(net.drop_p) if net.training else 0.0
    if planes is not None:
        m.lin_w_planes, m.lin_w_plane, m.lin_w_ld = planes.p.data_ptr(), planes.plane, planes.p.shape[2]
    else:
        m.lin_w_planes, m.lin_w_plane, m.lin_w_ld = None, 0, 0
    return m


def _batch_struct(prep):
    """RenetStepBatch of a prepared merged batch, built once and cached on it (with the tensors it points into)."""
    ent = getattr(prep, '_step_batch', None)
    if ent is not None:
        return ent[0]
    g = prep.g
    b = StepBatchC()
    it_src_t, it_type_t, col_t, e_src_t = g.table_items()
    heavy, heavy_out = g.heavy_rows, getattr(g, 'heavy_rows_out', None)
    vals = dict(N=g.N, E=g.E, nA=g.nA, S=g.S, B=prep.b, L=g.L, n_items=g.it_src.numel(), n_groups=g.n_groups,
                n_groups_out=g.n_groups_out, n_heavy=heavy.numel() if heavy is not None else 0,
                n_heavy_out=heavy_out.numel() if heavy_out is not None else 0, n_chunks=g.n_chunks, n_chunks2=g.n_chunks2)
    for n, v in vals.items():
        setattr(b, n, int(v))
    tens = dict(node_ent=g.node_ent, row_ptr=g.row_ptr, col=g.col, etype=g.etype, norm=g.norm, it_src=g.it_src,
                it_type=g.it_type, grp_ptr=g.grp_ptr, heavy_rows=heavy, heavy_rows_out=heavy_out, it_src_t=it_src_t,
                it_type_t=it_type_t, col_t=col_t, e_src_t=e_src_t, e_src=g.e_src, e_dst=g.e_dst, chunk_ptr=g.chunk_ptr,
                chunk_type=g.chunk_type, type_chunk_ptr=g.type_chunk_ptr, e_src2=g.e_src2, e_dst2=g.e_dst2,
                chunk_ptr2=g.chunk_ptr2, chunk_type2=g.chunk_type2, type_chunk_ptr2=g.type_chunk_ptr2, subj_row=g.subj_row,
                row_ent=g.row_ent, row_rel=g.row_rel, glob_row=g.glob_row, step_off=g.step_off, s_idx=prep.s_idx,
                r_idx=prep.r_idx, ent_label=prep.o_idx, rel_label=prep.r_label)
    for n, t in tens.items():
        if t is not None:
            if n == 'norm':
                K._f32(t, n)
            else:
                K._i32(t, n)
            setattr(b, n, t.data_ptr())
    b.step_off_host = ctypes.cast(prep.step_off, _P).value
    plans = []
    for n in ('plan_node_ent', 'plan_subj_row', 'plan_s', 'plan_r'):
        p = getattr(prep, n, None) if n in ('plan_s', 'plan_r') else getattr(g, n)
        if p is None:
            p = getattr(g, n)
        c = SegPlanC(K._i32(p.order), K._i32(p.seg_ptr), K._i32(p.target), int(p.num_segments))
        setattr(b, n, c)
        plans.append(p)
    prep._step_batch = (b, tens, plans, (it_src_t, it_type_t, col_t, e_src_t))
    return b


_events = {}
_hook_streams = {}


def _side_stream(device):
    """The side stream of ops._Side for this device, WITHOUT ordering it behind the current stream (the C side forks with its
    own events); None when side streams are off."""
    if not (ops.SIDE_STREAM and device.type == 'cuda'):
        return None
    key = device.index if device.index is not None else torch.cuda.current_device()
    st = ops._side_streams.get(key)
    if st is None:
        st = ops._side_streams[key] = torch.cuda.Stream(device=device)
    return st


def _head_event(device, which=0):
    """(event, stream) pair `which` of this device: 0 = the score head's gradients, 1 = the encoders' (the C side records
    the event; the reducer's hooks run on the stream, which waits for nothing but that event)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), which)
    ev = _events.get(key)
    if ev is None:
        ev = torch.cuda.Event()
        ev.record()                                  # (materialises the HIP event: the C side re-records it)
        _events[key] = ev
        _hook_streams[key] = torch.cuda.Stream(device=device)
    return ev, _hook_streams[key]


class StepFn(Function):
    """apply(net, prep, row_tap, *model_params(net)) -> the training loss of loss_prepared_both(prep)."""

    @staticmethod
    def forward(ctx, net, prep, row_tap, *params):
        dev = net.ent_embeds.device
        g = prep.g
        p = float(net.drop_p) if net.training else 0.0
        # the dropout seeds, drawn in the order of the autograd path (Aggregator.encode: rgcn1, rgcn2 [graph sites],
        # X, Xr; then the two heads)
        with ops.shared_graph_seeds(getattr(prep, 'sharded', False) or ops.SHARED_GRAPH_SEEDS):
            s_r1 = ops.next_seed(graph_site=True) if p > 0 else 0
            s_r2 = ops.next_seed(graph_site=True) if p > 0 else 0
            s_x, s_xr = (ops.next_seed(), ops.next_seed()) if p > 0 else (0, 0)
        s_h1, s_h2 = (ops.next_seed(), ops.next_seed()) if p > 0 else (0, 0)
        planes = K.weight_planes(params[14]) if K.use_planes(params[14], params[14].shape[0]) else None
        m = _model_struct(net, params, planes)
        m.glob = K._f32(g.glob, 'global embeddings')
        b = _batch_struct(prep)
        lib = K.lib()
        nbytes = lib.renet_step_workspace(ctypes.addressof(m), ctypes.addressof(b))
        if nbytes == 0:
            raise K.RenetHipError('renet_step_workspace rejected the model / batch')
        ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        scale = 2.0 * getattr(prep, 'share', 1.0)
        bb = prep.b
        s1, s2 = float(scale) / bb, float(scale) * 0.1 / bb
        rl = torch.empty(2 * bb, device=dev, dtype=torch.float32)
        side = _side_stream(dev)
        r = StepRunC(int(s_r1), int(s_r2), int(s_x), int(s_xr), int(s_h1), int(s_h2), s1, s2, ws.data_ptr(), nbytes,
                     K._stream(), side.cuda_stream if side is not None else None)
        nl = ctypes.c_int(0)
        K._check(lib.renet_step_forward(ctypes.addressof(m), ctypes.addressof(b), ctypes.addressof(r), rl.data_ptr(),
                                        ctypes.addressof(nl)), 'renet_step_forward')
        ctx.net, ctx.prep, ctx.m, ctx.b, ctx.r, ctx.ws, ctx.planes = net, prep, m, b, r, ws, planes
        ctx.consumed = False
        ctx.launches = nl.value
        ctx.gemm_mode = K.current_mode()
        ctx.side = side
        net.aggregator.last_batch = g
        StepFn.last_launches = [nl.value, 0]
        if row_tap is not None:
            row_tap.append((rl.detach(), s1, s2))
        return torch.dot(rl, ops._loss_weight_vector(bb, s1, s2, dev))

    @staticmethod
    def backward(ctx, g):
        if ctx.consumed:
            raise RuntimeError('StepFn: the saved CE gradients were consumed by the first backward pass; a second pass over '
                               'the same graph is not supported')
        ctx.consumed = True
        params = model_params(ctx.net)
        m, b, r = ctx.m, ctx.b, ctx.r
        # gradient targets: the parameter's own .grad buffer (accumulated into, autograd gets None) or, where there is none,
        # a slice of one freshly zeroed flat buffer that is returned to autograd (which then sets .grad)
        missing = [i for i, p in enumerate(params) if not _grad_ok(p)]
        fresh = [None] * len(params)
        if missing:
            if any(params[i].grad is not None for i in missing):
                raise RuntimeError('StepFn: a parameter has an unusable .grad (non-contiguous / wrong dtype); set '
                                   'RENET_STEP_PLAN=0')
            offs, tot = [], 0
            for i in missing:
                offs.append(tot)
                tot += (params[i].numel() + 3) & ~3
            flat = torch.zeros(tot, device=params[0].device, dtype=torch.float32)
            for i, o in zip(missing, offs):
                fresh[i] = flat[o:o + params[i].numel()].view_as(params[i])
        for i, (n, p) in enumerate(zip(_PARAMS, params)):
            setattr(m, 'g_' + n, (fresh[i] if fresh[i] is not None else p.grad).data_ptr())
        dev = ctx.net.ent_embeds.device
        side = ctx.side
        r.stream = K._stream()
        r.side_stream = side.cuda_stream if side is not None else None
        defer = bool(ops.DEFER_WEIGHT_GRADS and side is not None and not missing)
        lin_w, lin_b = params[14], params[15]
        hooked = (ops.hooks_wanted(lin_w) or ops.hooks_wanted(lin_b)) and not missing
        ev = hs = ev2 = hs2 = None
        if hooked:
            ev, hs = _head_event(dev)
        gru_params = params[6:14]                    # encoder.*, encoder_r.* (the reducer's middle bucket)
        hooked_gru = not missing and any(ops.hooks_wanted(p) for p in gru_params)
        if hooked_gru:
            ev2, hs2 = _head_event(dev, 1)
        g = g.contiguous()
        nl = ctypes.c_int(0)
        lib = K.lib()
        K._check(lib.renet_step_backward(ctypes.addressof(m), ctypes.addressof(b), ctypes.addressof(r), K._f32(g),
                                         int(defer), ev.cuda_event if ev is not None else None,
                                         ev2.cuda_event if ev2 is not None else None, ctypes.addressof(nl)),
                 'renet_step_backward')
        StepFn.last_launches[1] = nl.value
        if hooked_gru:
            hs2.wait_event(ev2)
            with torch.cuda.stream(hs2):
                for p in gru_params:
                    ops.grad_done(p)
        if hooked:
            # the score head's gradients are complete at `ev`, early in the launch list: the reducer's all-reduce of that
            # bucket is ordered behind the event only (a stream that waited for nothing else), not behind the whole pass
            hs.wait_event(ev)
            with torch.cuda.stream(hs):
                ops.grad_done(lin_w)
                ops.grad_done(lin_b)
        keep = (ctx.ws, ctx.prep, ctx.planes, m, b, g)
        if defer:
            # the workspace (and everything else the un-joined side-stream kernels read) stays referenced until
            # ops.join_deferred() has ordered this stream behind the side stream: it is then freed in stream order.  (No
            # record_stream: a block marked as used by another stream is held back until that stream's event has COMPLETED,
            # and a launching thread that runs hundreds of steps ahead of the GPU then piles up one 1.2 GB workspace per
            # step -- measured: 3.6 ms of host time per step over 200 steps.)
            ops._deferred.append((torch.cuda.current_stream(dev), side, keep))
        ctx.ws = None
        return (None, None, None) + tuple(fresh)


StepFn.last_launches = [0, 0]
