"""ctypes binding of librenet_hip.so (include/renet_hip.h).  PyTorch-ROCm is only the owner of device
memory and streams here: every function takes torch CUDA(HIP) tensors, checks dtype/contiguity,
and passes raw device pointers + the current stream to the C ABI.

There is NO fallback: if the library is missing or a call fails, this raises.
"""
import contextlib
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('RENET_HIP_LIB', os.path.join(_HERE, 'csrc', 'librenet_hip.so'))   # env: A/B builds

_lib = None

c_int = ctypes.c_int
c_float = ctypes.c_float
c_void_p = ctypes.c_void_p
c_size_t = ctypes.c_size_t
c_u64 = ctypes.c_uint64

_SIGNATURES = {
    'renet_version': (c_int, []),
    'renet_gather_rows': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'renet_segment_add': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'renet_segment_add2': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                   c_void_p]),
    'renet_rgcn_gather': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                  c_int, c_int, c_void_p, c_float, c_u64, c_int, c_void_p, c_int, c_void_p,
                                  c_int, c_int, c_int, c_int, c_void_p]),
    'renet_rgcn_gather_items': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_float, c_u64,
                                        c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'renet_rgcn_gather_items_bf16': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                             c_float, c_u64, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int,
                                             c_int, c_void_p]),
    'renet_rgcn_gather_items_table_bf16': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                                   c_int, c_int, c_void_p, c_float, c_u64, c_int, c_void_p, c_int,
                                                   c_void_p, c_int, c_void_p]),
    'renet_rgcn_gather_items_table': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                              c_float, c_u64, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    'renet_compose_table_items': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_void_p]),
    'renet_rgcn_bwd_prep': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_float, c_u64, c_int, c_int,
                                    c_void_p, c_void_p, c_void_p]),
    'renet_rgcn_bwd_prep_bounds': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_float, c_u64, c_int, c_int,
                                           c_void_p, c_void_p, c_void_p, c_void_p]),
    'renet_bound_parts': (c_int, [c_size_t]),
    'renet_rgcn_bwd_w_workspace': (c_size_t, [c_int, c_int]),
    'renet_rgcn_bwd_w': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                 c_void_p, c_int, c_int, c_int, c_void_p, c_float, c_void_p, c_size_t, c_void_p]),
    'renet_rgcn_bwd_w64': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                 c_void_p, c_int, c_int, c_int, c_void_p, c_float, c_void_p, c_size_t, c_void_p]),
    'renet_gemm_workspace': (c_size_t, [c_int, c_int, c_int]),
    'renet_gemm_f32': (c_int, [c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int, c_void_p, c_int,
                               c_float, c_void_p, c_int, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    'renet_gemm_f32_split': (c_int, [c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int, c_void_p, c_int,
                                     c_float, c_void_p, c_int, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    'renet_gemm_bf16': (c_int, [c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int, c_void_p, c_int,
                                c_float, c_void_p, c_int, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    'renet_gemm_split_plan': (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    'renet_maxabs_blocks': (c_int, [c_int, c_int, c_int]),
    'renet_maxabs_partials_multi': (c_int, [c_void_p, c_int, c_void_p]),
    'renet_maxabs_partials': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'renet_gemm_f32_h3': (c_int, [c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int, c_void_p, c_int,
                                  c_float, c_void_p, c_int, c_void_p, c_int, c_void_p, c_size_t, c_void_p, c_int,
                                  c_void_p, c_int, c_void_p]),
    'renet_bf16_bytes': (c_size_t, [c_int, c_int]),
    'renet_pack_bf16': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'renet_gemm_bf16s': (c_int, [c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int, c_void_p, c_int,
                                 c_float, c_void_p, c_int, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    'renet_bf16_zero_padding': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'renet_seq_assemble_fwd_bf16': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_void_p, c_int, c_int, c_float, c_u64, c_u64, c_void_p, c_int, c_void_p,
                                            c_int, c_int, c_void_p]),
    'renet_softmax_ce_bf16': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_int, c_int,
                                      c_void_p]),
    'renet_gru_bwd_layouts_bf16out': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    'renet_colsum_bf16': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_float, c_void_p, c_size_t, c_void_p]),
    'renet_scale_bf16_by_device_scalar': (c_int, [c_void_p, c_size_t, c_void_p, c_void_p]),
    'renet_planes_elems': (c_size_t, [c_int, c_int]),
    'renet_pack_planes': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'renet_gemm_planes': (c_int, [c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_int, c_size_t,
                                  c_void_p, c_int, c_size_t, c_float, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                  c_void_p, c_size_t, c_void_p]),
    'renet_softmax_ce_planes': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_size_t,
                                        c_int, c_int, c_void_p]),
    'renet_step_workspace': (c_size_t, [c_void_p, c_void_p]),
    'renet_step_forward': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'renet_step_backward': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    'renet_add_inplace': (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    'renet_zero': (c_int, [c_void_p, c_size_t, c_void_p]),
    'renet_colsum_workspace': (c_size_t, [c_int, c_int]),
    'renet_colsum': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_float, c_void_p, c_size_t, c_void_p]),
    'renet_scale_by_device_scalar': (c_int, [c_void_p, c_size_t, c_void_p, c_void_p]),
    'renet_scale_by_device_scalar_bound': (c_int, [c_void_p, c_size_t, c_void_p, c_float, c_void_p, c_void_p]),
    'renet_seq_assemble_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_int, c_int, c_float, c_u64, c_u64, c_void_p, c_void_p,
                                       c_void_p]),
    'renet_seq_assemble_fwd_bounds': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_int, c_int, c_float, c_u64, c_u64, c_void_p, c_void_p,
                                              c_void_p, c_void_p, c_void_p]),
    'renet_seq_assemble_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float,
                                       c_u64, c_u64, c_void_p, c_void_p, c_void_p, c_void_p]),
    'renet_gru_workspace': (c_size_t, [c_int, c_int]),
    'renet_gru_fwd': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                              c_void_p, c_size_t, c_void_p]),
    'renet_gru_bwd': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_size_t, c_void_p]),
    'renet_gru_fwd_multi': (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                    c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    'renet_gru_bwd_multi': (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_size_t, c_void_p]),
    'renet_gru_fwd_layouts': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'renet_gru_bwd_layouts': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_size_t, c_void_p]),
    'renet_gru_bound_parts': (c_int, [c_int]),
    'renet_gru_bwd_layouts_bounds': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'renet_gru_fwd_layouts_f32': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'renet_gru_bwd_layouts_f32': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_size_t, c_void_p]),
    'renet_gru_fwd_layouts_bf16': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'renet_gru_bwd_layouts_bf16': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_size_t, c_void_p]),
    'renet_concat3_fwd': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float,
                                  c_u64, c_void_p, c_void_p]),
    'renet_concat3_fwd_bounds': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float,
                                         c_u64, c_void_p, c_void_p, c_void_p]),
    'renet_concat3_bwd': (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_u64, c_void_p, c_void_p,
                                  c_void_p, c_void_p]),
    'renet_dropout': (c_int, [c_void_p, c_size_t, c_float, c_u64, c_void_p, c_void_p]),
    'renet_softmax_ce': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p,
                                 c_void_p]),
    'renet_joint_softmax': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    'renet_topk_workspace': (c_size_t, [c_int]),
    'renet_topk_positive': (c_int, [c_void_p, c_size_t, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t,
                                    c_void_p]),
    'renet_build_batch_workspace': (c_size_t, [c_void_p, c_int, c_int, c_int]),
    'renet_build_batch_both': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t,
                                       c_void_p]),
    'renet_adam_workspace': (c_size_t, [c_size_t]),
    'renet_adam_step': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_float, c_float, c_float,
                                c_float, c_float, c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    'renet_adam_step_scaled': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_float, c_float,
                                       c_float, c_float, c_float, c_float, c_int, c_int, c_void_p, c_size_t, c_void_p,
                                       c_void_p]),
    'renet_sumsq_partials': (c_int, [c_void_p, c_size_t, c_void_p, c_int, c_void_p]),
    'renet_adam_step_presummed': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_float, c_float,
                                          c_float, c_float, c_float, c_float, c_int, c_int, c_void_p, c_int, c_void_p,
                                          c_void_p]),
    'renet_host_filter_edges': (ctypes.c_int64, [c_void_p] * 5 + [ctypes.c_int64, ctypes.c_int64, c_void_p, c_void_p,
                                                 ctypes.c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    'renet_host_filter_edges_sparse': (ctypes.c_int64, [c_void_p] * 7 + [ctypes.c_int64, ctypes.c_int64, c_void_p,
                                                        c_void_p, ctypes.c_int64, c_void_p, c_void_p, c_void_p,
                                                        c_void_p]),
    'renet_host_node_sets': (ctypes.c_int64, [ctypes.c_int64] + [c_void_p] * 5 + [ctypes.c_int64, ctypes.c_int64]
                             + [c_void_p] * 7),
    'renet_host_type_chunks': (ctypes.c_int64, [ctypes.c_int64, c_void_p, c_void_p, c_void_p, ctypes.c_int64,
                                                ctypes.c_int64, ctypes.c_int64] + [c_void_p] * 6),
    'renet_host_edge_layouts': (None, [ctypes.c_int64, ctypes.c_int64, c_void_p, c_void_p, c_void_p, ctypes.c_int64,
                                       ctypes.c_int64, ctypes.c_int64] + [c_void_p] * 12),
    'renet_host_segplan': (ctypes.c_int64, [c_void_p, ctypes.c_int64, ctypes.c_int64, c_void_p, c_void_p, c_void_p]),
    'renet_host_gather_items': (ctypes.c_int64, [ctypes.c_int64, c_void_p, c_void_p, c_void_p, ctypes.c_int64,
                                                 ctypes.c_int64, ctypes.c_int64] + [c_void_p] * 5),
    'renet_segment_pool_fwd': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'renet_segment_pool_bwd': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                       c_void_p]),
}

EXPORTS = sorted(_SIGNATURES)


class RenetHipError(RuntimeError):
    pass


class _Bound(object):
    """The bound entry points (attribute per symbol)."""


def _exact_arity(fn, name, n):
    # ctypes silently accepts EXTRA positional arguments and converts them as 32-bit ints (a pointer passed
    # that way is truncated): insist on the declared arity
    def call(*a):
        if len(a) != n:
            raise RenetHipError('%s takes %d arguments, %d given' % (name, n, len(a)))
        return fn(*a)
    call.__name__ = name
    return call


def lib():
    """Loads librenet_hip.so (once).  Raises if it has not been built -- no CPU/torch fallback."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise RenetHipError('librenet_hip.so not built (%s); run `python re-net_amd/build.py` '
                                '-- there is no fallback path' % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        ns = _Bound()
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)       # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
            setattr(ns, name, _exact_arity(fn, name, len(args)))
        if ns.renet_version() != 1:
            raise RenetHipError('ABI version mismatch')
        ns._cdll = L
        _lib = ns
    return _lib


def _check(rc, what):
    if rc != 0:
        raise RenetHipError('%s failed with code %d' % (what, rc))


def _f32(t, name='tensor'):
    if t is None:
        return None
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise RenetHipError('%s must be a contiguous float32 device tensor' % name)
    return t.data_ptr()


def _i32(t, name='index'):
    if t is None:
        return None
    if not (t.is_cuda and t.dtype == torch.int32 and t.is_contiguous()):
        raise RenetHipError('%s must be a contiguous int32 device tensor' % name)
    return t.data_ptr()


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream():
    """Handle of torch's CURRENT stream on the current device (the stream every C-ABI call launches on).  Through the raw
    accessor when this torch has it: `torch.cuda.current_stream()` builds a Python Stream object per call (~12 us, 50+ times
    per training step: a quarter of the thread time a step needs to enqueue, tools/host_profile.py)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


# ---- optional per-kernel-class timing with HIP events on the launch stream (bench.py) -----------
class KernelTimer(object):
    """Collects (start, stop) HIP events around C-ABI calls plus the ALGORITHMIC work of each call.
    torch.cuda.Event records on torch's current stream, which is the stream the call launches on."""

    def __init__(self):
        self.recs = {}

    def begin(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def end(self, name, e0, flops=0.0, nbytes=0.0, tag=None):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.recs.setdefault(name, []).append((e0, e1, flops, nbytes, tag))

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, rs in self.recs.items():
            out[name] = {'calls': len(rs), 'ms': sum(r[0].elapsed_time(r[1]) for r in rs),
                         'flops': sum(r[2] for r in rs), 'bytes': sum(r[3] for r in rs)}
        return out

    def by_tag(self, name):
        """per-tag breakdown of one class (GEMMs: tag = (ta, tb, m, n, k, split_k))"""
        torch.cuda.synchronize()
        out = {}
        for r in self.recs.get(name, []):
            o = out.setdefault(r[4], {'calls': 0, 'ms': 0.0, 'flops': 0.0})
            o['calls'] += 1
            o['ms'] += r[0].elapsed_time(r[1])
            o['flops'] += r[2]
        return out


_timer = None

# wrappers without a timer of their own (the step's small kernels): set_timer(t, glue=True) times each of them as class
# 'glue:<name>' so that bench.py's `kernel_only` covers EVERY C-ABI launch of a step, not only the named classes
GLUE = ('softmax_ce_planes', 'gather_rows', 'segment_add', 'segment_add2', 'compose_table_items', 'rgcn_bwd_prep', 'colsum',
        'scale_by_device_scalar', 'seq_assemble_fwd', 'seq_assemble_fwd_bf16', 'softmax_ce_bf16', 'seq_assemble_bwd',
        'concat3_fwd', 'concat3_bwd', 'dropout', 'softmax_ce', 'adam_step', 'segment_pool_fwd', 'segment_pool_bwd',
        'pack_bf16_glue')
_glue_saved = {}


def _timed_glue(name, fn):
    import functools

    @functools.wraps(fn)
    def call(*a, **k):
        t = _timer
        if t is None:
            return fn(*a, **k)
        e0 = t.begin()
        try:
            return fn(*a, **k)
        finally:
            t.end('glue:' + name, e0)
    return call


def set_timer(t, glue=False):
    """Installs (None: removes) the kernel timer.  glue=True also times the wrappers listed in GLUE."""
    global _timer
    _timer = t
    g = globals()
    for name, fn in _glue_saved.items():
        g[name] = fn
    _glue_saved.clear()
    if t is not None and glue:
        for name in GLUE:
            if name in g and callable(g[name]):
                _glue_saved[name] = g[name]
                g[name] = _timed_glue(name, g[name])


# ---- thin typed wrappers -----------------------------------------------------------------------
def gather_rows(table, idx, out=None):
    n, d = idx.numel(), table.shape[1]
    if out is None:
        out = torch.empty(n, d, device=table.device, dtype=torch.float32)
    _check(lib().renet_gather_rows(_f32(table), _i32(idx), n, d, _f32(out), _stream()), 'gather_rows')
    return out


def segment_add(src, plan, dst):
    """dst[plan.target[u]] += sum of src rows listed in plan (deterministic)."""
    _check(lib().renet_segment_add(_f32(src), _i32(plan.order), _i32(plan.seg_ptr), _i32(plan.target),
                                   plan.num_segments, src.shape[1], _f32(dst), _stream()), 'segment_add')
    return dst


def segment_add2(src0, src1, plan, dst0, dst1):
    """dst0[target] += segment sums of src0 and dst1[target] += segment sums of src1, one launch, one plan."""
    _check(lib().renet_segment_add2(_f32(src0), _f32(src1), _i32(plan.order), _i32(plan.seg_ptr), _i32(plan.target),
                                    plan.num_segments, src0.shape[1], _f32(dst0), _f32(dst1), _stream()), 'segment_add2')


def rgcn_gather(x, row_ptr, col, etype, scale, weight, type_shift, transpose_w, addend, drop_p, seed,
                relu, out, heavy_rows=None, heavy_thresh=0, src_limit=0, addend_rows=0, n_edges=None):
    """n_edges: edges actually walked by this launch (pruned launches), for the byte accounting only."""
    n, d = x.shape[0], x.shape[1]
    t0 = _timer.begin() if _timer is not None else None
    _check(lib().renet_rgcn_gather(_f32(x), d, _i32(row_ptr), _i32(col), _i32(etype), _f32(scale),
                                   _f32(weight), weight.shape[0], type_shift, int(transpose_w),
                                   _f32(addend), float(drop_p), int(seed), int(relu), _f32(out),
                                   out.shape[0], _i32(heavy_rows) if heavy_rows is not None else None,
                                   heavy_rows.numel() if heavy_rows is not None else 0, int(heavy_thresh),
                                   int(src_limit), int(addend_rows), _stream()), 'rgcn_gather')
    if t0 is not None:
        # algorithmic bytes: per edge one source row + src + type index; per node one output
        # row + row_ptr + norm (+ the fused addend row); the relation weight table once
        nn = out.shape[0]
        e = col.numel() if n_edges is None else int(n_edges)
        nbytes = e * (d * 4 + 8) + nn * (d * 4 + 8) + weight.numel() * 4 + (nn * d * 4 if addend is not None else 0)
        # launches over the full batch graph (layer 1 and its backward) and the pruned ones (last layer:
        # subject rows only, a few thousand short rows -- launch-latency bound) are different regimes
        _timer.end('rgcn_gather' if n_edges is None else 'rgcn_gather_pruned', t0, nbytes=float(nbytes))
    return out


def gather_bytes(n_edges, n_rows, d, weight_numel, has_addend, x_bytes=4, w_bytes=4):
    """Algorithmic bytes of one gather-SpMM launch: per edge one source row + its source and type
    index; per output row the row itself + row_ptr + norm (+ the fused self-loop addend row, which the formula
    allows to count when the epilogue is fused); the relation weight table once.  x_bytes / w_bytes: element size
    of the source rows / the relation table as STORED (2 in the bf16-storage forms)."""
    return (n_edges * (d * x_bytes + 8) + n_rows * (d * 4 + 8) + weight_numel * w_bytes +
            (n_rows * d * 4 if has_addend else 0))


def rgcn_gather_items(x, g, weight, type_shift, transpose_w, addend, drop_p, seed, relu, out, use_norm=True,
                      pruned=False, src_limit=0, addend_rows=0, w16=None):
    """renet_rgcn_gather_items on the planned item stream of DeviceGraph `g` (graph.plan_gather_items).
    pruned: the launch covers the row prefix [0, out.shape[0]) = g.nA (groups / hub rows of that prefix) in the
    forward, or all rows with the edges whose source is >= src_limit skipped in the backward.
    w16: the bf16 copy of `weight` (gather_weight_bf16; bf16-storage mode) -- the relation blocks are then read as bf16."""
    d = x.shape[1]
    n_rows = out.shape[0]
    prefix = n_rows < g.N
    n_groups = g.n_groups_out if prefix else g.n_groups
    heavy = g.heavy_rows_out if prefix else g.heavy_rows
    if max(x.numel(), out.numel()) * 4 >= (1 << 31):
        # the item kernels address with 32-bit buffer offsets (< 2 GiB per tensor: 2.6 M rows at D = 200); beyond
        # that the plain-CSR kernel with 64-bit addressing takes over
        return rgcn_gather(x, g.row_ptr, g.col, g.etype, g.norm if use_norm else None, weight, type_shift,
                           transpose_w, addend, drop_p, seed, relu, out, heavy, g.heavy_thresh, src_limit,
                           addend_rows, n_edges=g.E_out if pruned else None)
    t0 = _timer.begin() if _timer is not None else None
    hv, nhv = (_i32(heavy), heavy.numel()) if heavy is not None else (None, 0)
    if w16 is not None:
        _check(lib().renet_rgcn_gather_items_bf16(
            _f32(x), d, _i32(g.it_src), _i32(g.it_type), _i32(g.grp_ptr), int(n_groups), _i32(g.row_ptr), _i32(g.col),
            _i32(g.etype), _f32(g.norm) if use_norm else None, w16.p.data_ptr(), w16.p.shape[1], weight.shape[0],
            int(type_shift), int(transpose_w), _f32(addend), float(drop_p), int(seed), int(relu), _f32(out), n_rows,
            hv, nhv, int(src_limit), int(addend_rows), int(pruned), _stream()), 'rgcn_gather_items_bf16')
    else:
        _check(lib().renet_rgcn_gather_items(
            _f32(x), d, _i32(g.it_src), _i32(g.it_type), _i32(g.grp_ptr), int(n_groups), _i32(g.row_ptr), _i32(g.col),
            _i32(g.etype), _f32(g.norm) if use_norm else None, _f32(weight), weight.shape[0], int(type_shift),
            int(transpose_w), _f32(addend), float(drop_p), int(seed), int(relu), _f32(out), n_rows, hv, nhv,
            int(src_limit), int(addend_rows), int(pruned), _stream()), 'rgcn_gather_items')
    if t0 is not None:
        e = g.E_out if pruned else g.E
        name = 'rgcn_gather_%s_%s' % ('bwdh' if transpose_w else 'fwd', 'pruned' if pruned else 'full')
        wb = 2 if w16 is not None else 4
        _timer.end(name, t0, nbytes=float(gather_bytes(e, n_rows, d, weight.numel(), addend is not None, 4, wb)),
                   tag=float(gather_bytes(e, n_rows, d, weight.numel(), False, 4, wb)))       # tag: the strict bytes
    return out


def gather_weight_bf16(weight):
    """bf16 copy of an fp32 matrix the gather kernels read (the relation-block table, the entity table) in
    bf16-storage mode, else None.  Registered weights come from the per-optimizer-step cache (_as_bf16).
    RENET_BF16_GATHER=0 keeps the gathers on fp32 operands."""
    if current_mode() != 'bf16s' or os.environ.get('RENET_BF16_GATHER', '1') == '0':
        return None
    if weight.numel() * 2 >= (1 << 31):
        return None
    return _as_bf16(weight)[0]


def compose_table_items(g):
    """Index arrays of the table-addressed first layer for DeviceGraph `g` (renet_compose_table_items): source rows
    composed through g.node_ent.  -> (it_src_t, it_type_t, col_t, e_src_t), int32 device tensors."""
    ni, e = g.it_src.numel(), g.col.numel()
    buf = torch.empty(2 * ni + 2 * e + 16, device=g.col.device, dtype=torch.int32)
    a0, a1, a2 = (ni + 3) & ~3, 2 * ((ni + 3) & ~3), 2 * ((ni + 3) & ~3) + ((e + 3) & ~3)
    it_src_t, it_type_t, col_t, e_src_t = buf[:ni], buf[a0:a0 + ni], buf[a1:a1 + e], buf[a2:a2 + e]
    _check(lib().renet_compose_table_items(_i32(g.node_ent), _i32(g.it_src), _i32(g.it_type), ni, _i32(g.col),
                                           _i32(g.e_src), e, it_src_t.data_ptr(), it_type_t.data_ptr(),
                                           col_t.data_ptr(), e_src_t.data_ptr(), _stream()), 'compose_table_items')
    return it_src_t, it_type_t, col_t, e_src_t


def rgcn_gather_items_table(table, g, weight, type_shift, addend_table, drop_p, seed, relu, out, table16=None,
                            w16=None):
    """renet_rgcn_gather_items_table: the first RGCN layer reading source rows and the self-loop addend through
    g.node_ent from [N_ent, D] tables (forward, full graph).  table16 + w16 (gather_weight_bf16): source rows and
    relation blocks are read from the bf16 copies (the addend table stays fp32)."""
    d = table.shape[1]
    n_rows = out.shape[0]
    it_src_t, it_type_t, col_t, _ = g.table_items()
    heavy = g.heavy_rows
    t0 = _timer.begin() if _timer is not None else None
    if table16 is not None and w16 is not None:
        _check(lib().renet_rgcn_gather_items_table_bf16(
            table16.p.data_ptr(), table16.p.shape[1], table.shape[0], d, _i32(it_src_t), _i32(it_type_t),
            _i32(g.grp_ptr), int(g.n_groups), _i32(g.row_ptr), _i32(col_t), _i32(g.etype), _i32(g.node_ent),
            _f32(g.norm), w16.p.data_ptr(), w16.p.shape[1], weight.shape[0], int(type_shift), _f32(addend_table),
            float(drop_p), int(seed), int(relu), _f32(out), n_rows, _i32(heavy) if heavy is not None else None,
            heavy.numel() if heavy is not None else 0, _stream()), 'rgcn_gather_items_table_bf16')
    else:
        _check(lib().renet_rgcn_gather_items_table(
            _f32(table), table.shape[0], d, _i32(it_src_t), _i32(it_type_t), _i32(g.grp_ptr), int(g.n_groups),
            _i32(g.row_ptr), _i32(col_t), _i32(g.etype), _i32(g.node_ent), _f32(g.norm), _f32(weight), weight.shape[0],
            int(type_shift), _f32(addend_table), float(drop_p), int(seed), int(relu), _f32(out), n_rows,
            _i32(heavy) if heavy is not None else None, heavy.numel() if heavy is not None else 0, _stream()),
            'rgcn_gather_items_table')
    if t0 is not None:          # SURVEY 8d's ALGORITHMIC bytes: one source row per edge, one output (+ addend) row per node
        eb = 2 if (table16 is not None and w16 is not None) else 4
        _timer.end('rgcn_gather_fwd_full', t0,
                   nbytes=float(gather_bytes(g.E, n_rows, d, weight.numel(), addend_table is not None, eb, eb)),
                   tag=float(gather_bytes(g.E, n_rows, d, weight.numel(), False, eb, eb)))
    return out


def rgcn_bwd_prep(g_out, out, norm, relu, drop_p, seed, gn, g_loop):
    n, d = g_out.shape
    if _fused_bounds(g_loop):                    # g_loop's operand bound comes out of the kernel that writes it
        nparts = lib().renet_bound_parts(n * (d // 4))
        part = torch.empty(nparts, device=g_loop.device, dtype=torch.float32)
        _check(lib().renet_rgcn_bwd_prep_bounds(_f32(g_out), _f32(out), _f32(norm), int(relu), float(drop_p),
                                                int(seed), n, d, _f32(gn), _f32(g_loop), part.data_ptr(), _stream()),
               'rgcn_bwd_prep_bounds')
        _note_bound(g_loop, part, nparts)
        return
    _check(lib().renet_rgcn_bwd_prep(_f32(g_out), _f32(out), _f32(norm), int(relu), float(drop_p),
                                     int(seed), n, d, _f32(gn), _f32(g_loop), _stream()), 'rgcn_bwd_prep')


def rgcn_bwd_w(x, gn, e_src, e_dst, chunk_ptr, chunk_type, n_chunks, type_chunk_ptr, num_types, type_shift,
               dW, beta=0.0):
    d = x.shape[1]
    # feature tensors of 2 GiB and more: the entry with 64-bit addressing (as rgcn_gather_items falls back to the CSR
    # kernel); RENET_BWDW_64=1 forces it (tests)
    big = max(x.numel(), gn.numel()) * 4 >= (1 << 31) or os.environ.get('RENET_BWDW_64') == '1'
    nbytes = lib().renet_rgcn_bwd_w_workspace(n_chunks, d)
    ws = torch.empty(max(nbytes // 4, 1), device=x.device, dtype=torch.float32)
    t0 = _timer.begin() if _timer is not None else None
    fn = lib().renet_rgcn_bwd_w64 if big else lib().renet_rgcn_bwd_w
    _check(fn(_f32(x), _f32(gn), _i32(e_src), _i32(e_dst), _i32(chunk_ptr), _i32(chunk_type), n_chunks,
              _i32(type_chunk_ptr), num_types, type_shift, d, _f32(dW), float(beta), ws.data_ptr(), nbytes, _stream()),
           'rgcn_bwd_w')
    if t0 is not None:      # SURVEY 8d backward-W: E * 2 rows + indices, dW written once
        _timer.end('rgcn_bwd_w', t0, nbytes=float(e_src.numel() * (2 * d * 4 + 8) + dW.numel() * 4))
    return dW


def _ld(t):
    if t.dim() != 2 or t.stride(1) != 1:
        raise RenetHipError('GEMM operands must be 2-D with unit inner stride')
    return t.stride(0)


def gemm_split_plan(ta, tb, m, n, k, lda=None, ldb=None, split_k=1, a_ptr=0, b_ptr=0):
    """What renet_gemm_f32_split would launch for this problem (host logic only, no GPU needed): dict with `kernel`
    ('fused' | 'two_phase_128' | 'two_phase_256' | 'weight_resident'), `raw` (raw-buffer loader), `xcd_order` (0 = plain
    tile order, w = XCD-aware order with panels of <= w tiles), `grid`, `split_k`."""
    import ctypes
    lda = (m if ta else k) if lda is None else lda
    ldb = (k if tb else n) if ldb is None else ldb
    out = (ctypes.c_int * 8)()
    _check(lib().renet_gemm_split_plan(int(ta), int(tb), m, n, k, a_ptr, lda, b_ptr, ldb, split_k,
                                       ctypes.cast(out, ctypes.c_void_p)), 'renet_gemm_split_plan')
    names = {0: 'fused', 1: 'two_phase_128', 2: 'two_phase_256', 3: 'weight_resident', 4: 'bf16'}
    return {'kernel': names[out[0]], 'raw': bool(out[1]), 'xcd_order': out[2], 'grid': (out[3], out[4], out[5]),
            'split_k': out[6]}


def auto_split_k(m, n, k):
    """Deterministic split-K factor for outputs with too few 128x128 tiles to fill the chip (weight-gradient and
    dX-of-the-head shapes), from a small cost model in units of one k-tile step of a workgroup (~1.5 us):
        T(s) = rounds(tiles * s / 512 slots) * (ktiles / s + 6 [prologue + C store]) + s * M*N * 2.7e-6 [partial
        sums written and re-read by the reduce kernel]
    It reproduces the factors swept on MI355X (tools/gemm_sweep.py: dfeat 40 tiles -> 12 slices, dW_ih 35 -> 14,
    dW_loop 4 -> 128) and, unlike a pure fill heuristic, still splits a single-tile output (n_hidden = 100:
    100x100xK=46k would otherwise run 1440 k-tiles in ONE workgroup)."""
    tiles = ((m + 127) // 128) * ((n + 127) // 128)
    ktiles = (k + 31) // 32
    if tiles >= 256 or ktiles < 8:
        return 1
    smax = int(max(1, min(128, ktiles // 4)))
    per_slice = m * n * 2.7e-6
    best, best_t = 1, None
    for s in range(1, smax + 1):
        rounds = (tiles * s + 511) // 512
        t = rounds * (ktiles / float(s) + 6.0) + (per_slice * s if s > 1 else 0.0)
        if best_t is None or t < best_t - 1e-9:
            best, best_t = s, t
    return best


# 'bf16s'  : bf16 STORAGE: operands are (packed to) bf16 matrices in HBM, LDS-DMA staging, one bf16 MFMA product
#            (gemm_bf16s; BASELINE config 5 -- bench.py --dtype bf16)
# 'bf16'   : operands rounded to bf16 on the way into LDS, ONE v_mfma_f32_32x32x16_bf16 product, fp32 accumulate
#            (mixed precision for BASELINE config 5; tensors stay fp32 in memory)
# 'f32'    : v_mfma_f32_32x32x2_f32, exact fp32 products (gemm.hip)
# 'bf16x6' : fp32 operands split into 3 bf16 terms, 6 term products on v_mfma_f32_32x32x16_bf16 with fp32
#            accumulation (gemm_split.hip): fp32-class accuracy at 2.67x the matrix-pipe rate
# 'f16x3'  : fp32 operands, scaled per TENSOR by a power of two, split into 2 binary16 terms (22 significant bits),
#            3 term products on v_mfma_f32_32x32x16_f16 (gemm_h3.h): GEMM errors within a small multiple of fp32's
#            own accumulation error (include/renet_hip.h), half the matrix instructions of bf16x6
# Default since round 4: 'bf16x6', the 24-bit split (fp32-class: every operand bit enters the product).  'f16x3' (22-bit
# operands, the round-3 default) is the opt-in FAST mode, 'f32' the exact-product mode; bench.py reports all three.
GEMM_MODE = os.environ.get('RENET_GEMM', 'bf16x6')

# Per-MODEL choice of the fp32-storage modes (round 4: the two split modes; round 5: 'f32' too): a model may carry
# `gemm_mode = 'f16x3' | 'bf16x6' | 'f32'` and its forward / backward passes then run inside `with gemm_mode(...)`; everything
# below reads the mode through current_mode() -- the GEMM entry point, and for 'f32' the exact GRU entries
# renet_gru_{fwd,bwd}_layouts_f32 (until round 4 the library chose them from RENET_GEMM alone).  GEMM_MODE stays the PROCESS
# default; 'bf16s' (operand STORAGE: packed weights, bf16 activations) remains process-wide: gemm_mode() accepts it only when
# it equals the process default.  (In a RENET_GEMM=f32 process the recurrences stay exact whatever a model asks for.)
_mode_override = threading.local()       # .mode: per THREAD (a prefetch / evaluation thread must not inherit another thread's
                                         # scope; ops._fwd_mode / _bwd_mode re-enter the scope on the autograd thread explicitly)
_PER_MODEL_MODES = ('bf16x6', 'f16x3', 'f32')


def current_mode():
    return getattr(_mode_override, 'mode', None) or GEMM_MODE


@contextlib.contextmanager
def gemm_mode(mode):
    """Scope in which the fp32-class GEMMs run in `mode` (None: no change)."""
    if mode is None or mode == current_mode():
        yield
        return
    if mode not in _PER_MODEL_MODES or GEMM_MODE not in _PER_MODEL_MODES:
        raise RenetHipError('gemm_mode(%r): only %s can be chosen per model, and only in a process whose default is one '
                            'of them (RENET_GEMM=%s)' % (mode, ' / '.join(_PER_MODEL_MODES), GEMM_MODE))
    old = getattr(_mode_override, 'mode', None)
    _mode_override.mode = mode
    try:
        yield
    finally:
        _mode_override.mode = old


class BF16Mat(object):
    """A matrix [R, C] stored as bf16 in a zero-padded [Rp, Cp] buffer (Rp, Cp multiples of 256): the operand format of
    gemm_bf16s (renet_pack_bf16 / a producer kernel that writes it directly)."""
    __slots__ = ('p', 'R', 'C')

    def __init__(self, p, R, C):
        self.p, self.R, self.C = p, R, C

    @property
    def shape(self):
        return (self.R, self.C)


def bf16_empty(r, c, device):
    """Uninitialised BF16Mat for a producer kernel that writes the bf16 format directly."""
    return BF16Mat(torch.empty((r + 255) & ~255, (c + 255) & ~255, device=device, dtype=torch.bfloat16), r, c)


def pack_bf16(x):
    """fp32 [R, C] (row-strided views allowed) -> BF16Mat."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1):
        raise RenetHipError('pack_bf16 needs a 2-D float32 device tensor with unit inner stride')
    r, c = x.shape
    rp, cp = (r + 255) & ~255, (c + 255) & ~255
    p = torch.empty(rp, cp, device=x.device, dtype=torch.bfloat16)
    t0 = _timer.begin() if _timer is not None else None
    _check(lib().renet_pack_bf16(x.data_ptr(), r, c, x.stride(0), p.data_ptr(), _stream()), 'pack_bf16')
    if t0 is not None:
        _timer.end('pack_bf16', t0, nbytes=float(r * c * 4 + rp * cp * 2))
    return BF16Mat(p, r, c)


# bf16 copies of WEIGHTS are cached between optimizer steps: parallel.HipAdam registers its parameters (stable
# addresses inside the flat buffer) and bumps the epoch after every update; any other tensor is packed on every use.
_weight_ptrs = {}              # data_ptr -> (rows, cols)
_weight_cache = {}             # data_ptr -> (epoch, BF16Mat)
_weight_epoch = [0]


def register_weights(tensors):
    import weakref
    for t in tensors:
        if t.dim() == 2 and t.is_contiguous():
            _weight_ptrs[t.data_ptr()] = tuple(t.shape)
            _weight_objs[t.data_ptr()] = weakref.ref(t)


def unregister_weights(tensors):
    for t in tensors:
        _weight_ptrs.pop(t.data_ptr(), None)
        _weight_cache.pop(t.data_ptr(), None)
        _weight_planes.pop(t.data_ptr(), None)
        _weight_max.pop(t.data_ptr(), None)
        _weight_objs.pop(t.data_ptr(), None)


def weights_changed():
    _weight_epoch[0] += 1


def _as_bf16(x):
    """Operand of gemm_bf16s for tensor / BF16Mat x -> (BF16Mat, rows, cols) of the LOGICAL matrix x."""
    if isinstance(x, BF16Mat):
        return x, x.R, x.C
    ptr = x.data_ptr()
    shp = _weight_ptrs.get(ptr)
    if shp is not None and x.dim() == 2 and x.stride(1) == 1 and x.stride(0) == shp[1] and \
            x.shape[0] <= shp[0] and x.shape[1] <= shp[1]:
        # a registered weight, or a leading row / column block of it (W_ih[:, :live]): one cached copy serves both
        ent = _weight_cache.get(ptr)
        stamp = (_weight_epoch[0], x._version)      # in-place torch writes (load_state_dict) bump _version
        if ent is None or ent[0] != stamp:
            full = torch.as_strided(x, shp, (shp[1], 1))
            ent = (stamp, pack_bf16(full))
            _weight_cache[ptr] = ent
        return ent[1], x.shape[0], x.shape[1]
    m = pack_bf16(x)
    return m, m.R, m.C


# ---- PLANES (round 6): operands of the bf16x6 GEMM stored ALREADY split into their three bf16 terms ----------------
# The entity score head's three GEMMs (logits, dfeat, dW: 2/3 of the step's GEMM flops) re-split every operand element
# once per output tile that reads it (csrc/gemm_split.hip); here a tensor is split ONCE, by the kernel that produces it
# (softmax_ce_planes: the CE gradient; HipAdam: the head weight; pack_planes: the features), and the GEMM
# (csrc/gemm_p6.h) stages the planes by LDS-DMA with no conversion work in its k-loop.  Same arithmetic (three RNE terms,
# six products, fp32 accumulation) as the in-loop split.  RENET_PLANES=0 keeps the in-loop split everywhere.
PLANES = os.environ.get('RENET_PLANES', '1') != '0'
PLANES_MIN_CLASSES = int(os.environ.get('RENET_PLANES_MIN_CLASSES', '2048'))


class PlanesMat(object):
    """A matrix [R, C] as three bf16 planes in ONE tensor p [3, Rp, Cp] (Rp, Cp multiples of 256, zero padding).  The
    elements of a plane are NOT row-major: they are stored in the T16 tile format of csrc/gemm_p6.h (16 x 16 tiles of 512
    bytes; csrc/common.h renet_t16_off, tools/p6_layout_sim.py) -- p's trailing two dimensions only size the buffer;
    planes_to_dense() gives the row-major view (tests)."""
    __slots__ = ('p', 'R', 'C')

    def __init__(self, p, R, C):
        self.p, self.R, self.C = p, R, C

    @property
    def shape(self):
        return (self.R, self.C)

    @property
    def plane(self):
        return self.p.shape[1] * self.p.shape[2]


def planes_to_dense(m):
    """-> float32 [3, Rp, Cp]: the planes of PlanesMat m in row-major order (undoes the T16 tiling; tests / debugging)."""
    rp, cp = m.p.shape[1], m.p.shape[2]
    dev = m.p.device
    r = torch.arange(rp, device=dev).view(-1, 1)
    c = torch.arange(cp, device=dev).view(1, -1)
    i, j, tc = r & 15, c & 15, c >> 4
    idx = ((r >> 4) * (cp // 16) + tc) * 256 + (i ^ ((tc & 1) << 2)) * 16 + ((j >> 3) ^ ((i >> 3) & 1)) * 8 + (j & 7)
    return m.p.reshape(3, -1)[:, idx.reshape(-1)].reshape(3, rp, cp).float()


def planes_empty(r, c, device, zero=False):
    """PlanesMat for a producer kernel; zero=True when the producer does not write the padding itself."""
    rp, cp = (r + 255) & ~255, (c + 255) & ~255
    mk = torch.zeros if zero else torch.empty
    return PlanesMat(mk(3, rp, cp, device=device, dtype=torch.bfloat16), r, c)


def pack_planes(x, ones_col=False):
    """fp32 [R, C] (row-strided views allowed) -> PlanesMat; ones_col: one more column, all ones (gemm_planes' col_out)."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1):
        raise RenetHipError('pack_planes needs a 2-D float32 device tensor with unit inner stride')
    r, c = x.shape
    m = planes_empty(r, c + (1 if ones_col else 0), x.device)
    t0 = _timer.begin() if _timer is not None else None
    _check(lib().renet_pack_planes(x.data_ptr(), r, c, x.stride(0), int(bool(ones_col)), m.p.data_ptr(), _stream()),
           'pack_planes')
    if t0 is not None:
        _timer.end('pack_planes', t0, nbytes=float(r * c * 4 + 3 * m.plane * 2))
    return m


_weight_planes = {}            # data_ptr -> ((epoch, version), PlanesMat) of a registered weight


def weight_planes(w):
    """Planes of a weight matrix: a registered weight (parallel.HipAdam) is packed once per optimizer step -- or not at
    all when the optimizer kernel wrote them (note_weight_planes) -- anything else on every call."""
    ptr = w.data_ptr()
    shp = _weight_ptrs.get(ptr)
    if shp is not None and tuple(w.shape) == tuple(shp) and w.is_contiguous():
        stamp = (_weight_epoch[0], w._version)
        ent = _weight_planes.get(ptr)
        if ent is None or ent[0] != stamp:
            ent = (stamp, pack_planes(w), _StreamGuard(w.device))
            _weight_planes[ptr] = ent
        ent[2].wait(w.device)
        return ent[1]
    return pack_planes(w)


def note_weight_planes(w, mat):
    """The optimizer kernel has just written `mat` = the planes of the UPDATED weight w (call after weights_changed())."""
    _weight_planes[w.data_ptr()] = ((_weight_epoch[0], w._version), mat, _StreamGuard(w.device))


def use_planes(weight, n_classes):
    """Whether a score head with this weight runs its GEMMs on planes: default fp32-class mode, a wide class dimension."""
    return (PLANES and current_mode() == 'bf16x6' and weight.is_cuda and n_classes >= PLANES_MIN_CLASSES and
            weight.dim() == 2 and weight.is_contiguous())


def auto_split_k_planes(m, n, k):
    """Split-K factor of gemm_planes (256 x 128 tiles, one workgroup per CU): auto_split_k's cost model on that grid."""
    tiles = ((m + 255) // 256) * ((n + 127) // 128)
    ktiles = (k + 31) // 32
    if tiles >= 128 or ktiles < 8:
        return 1
    smax = int(max(1, min(64, ktiles // 4)))
    per_slice = m * n * 2.7e-6
    best, best_t = 1, None
    for s_ in range(1, smax + 1):
        rounds = (tiles * s_ + 255) // 256
        t = rounds * (ktiles / float(s_) + 6.0) + (per_slice * s_ if s_ > 1 else 0.0)
        if best_t is None or t < best_t - 1e-9:
            best, best_t = s_, t
    return best


def gemm_planes(a, b, ta=False, tb=False, out=None, bias=None, alpha=1.0, alpha_dev=None, beta=0.0, col_out=None,
                split_k=None):
    """out = alpha * alpha_dev * op(a) @ op(b) + bias + beta * out on PlanesMat operands (renet_gemm_planes).
    ta: a is stored [K, M]; tb: b is stored [N, K] (the conventions of gemm()).  col_out [M]: b's LAST column (a ones
    column, pack_planes(ones_col=True)) is not part of `out`; its product -- sum_k op(a)[m, k] -- goes to col_out."""
    if not (isinstance(a, PlanesMat) and isinstance(b, PlanesMat)):
        raise RenetHipError('gemm_planes needs PlanesMat operands')
    m, k = (a.C, a.R) if ta else (a.R, a.C)
    n, k2 = (b.R, b.C) if tb else (b.C, b.R)
    if k != k2:
        raise RenetHipError('gemm_planes inner dimensions differ: %d vs %d' % (k, k2))
    n_main = n - 1 if col_out is not None else n
    dev = a.p.device
    if out is None:
        if beta != 0.0:
            raise RenetHipError('beta != 0 needs an output tensor')
        out = torch.empty(m, n_main, device=dev, dtype=torch.float32)
    if tuple(out.shape) != (m, n_main):
        raise RenetHipError('gemm_planes: output shape %r, expected %r' % (tuple(out.shape), (m, n_main)))
    if col_out is not None and (col_out.numel() != m or not col_out.is_contiguous()):
        raise RenetHipError('gemm_planes: col_out must be a contiguous vector of M floats')
    if split_k is None:
        split_k = auto_split_k_planes(m, n, k)
    ws_ptr, ws_bytes = None, 0
    if split_k > 1:
        ws_bytes = lib().renet_gemm_workspace(m, n, split_k)
        ws = torch.empty(ws_bytes // 4, device=dev, dtype=torch.float32)
        ws_ptr = ws.data_ptr()
    t0 = _timer.begin() if _timer is not None else None
    _check(lib().renet_gemm_planes(int(ta), int(not tb), m, n, k, float(alpha), _f32(alpha_dev), a.p.data_ptr(),
                                   a.p.shape[2], a.plane, b.p.data_ptr(), b.p.shape[2], b.plane, float(beta),
                                   out.data_ptr(), _ld(out), _f32(bias), _f32(col_out), split_k, ws_ptr, ws_bytes,
                                   _stream()), 'gemm_planes')
    if t0 is not None:
        _timer.end('gemm_f32', t0, flops=2.0 * m * n_main * k, tag=(int(ta), int(tb), m, n_main, k, split_k))
    return out


def softmax_ce_planes(logits, target, grad_scale, row_loss=None):
    """-> (row_loss[B], PlanesMat (softmax - onehot) * grad_scale); the fp32 logits (row-strided view allowed) are not
    modified."""
    b, c = logits.shape
    if row_loss is None:
        row_loss = torch.empty(b, device=logits.device, dtype=torch.float32)
    dl = planes_empty(b, c, logits.device)
    _check(lib().renet_softmax_ce_planes(logits.data_ptr(), _i32(target), b, c, _ld(logits), float(grad_scale),
                                         _f32(row_loss), dl.p.data_ptr(), dl.plane, dl.p.shape[2], dl.p.shape[1],
                                         _stream()), 'softmax_ce_planes')
    return row_loss, dl


class F32Op(object):
    """An fp32 GEMM operand together with the bound on its largest magnitude that the f16x3 GEMM scales it by:
    `part` holds n <= 256 device floats whose maximum bounds max |t| (renet_maxabs_partials, or a known bound)."""
    __slots__ = ('t', 'part', 'n')

    def __init__(self, t, part=None, n=0):
        self.t, self.part, self.n = t, part, n

    @property
    def shape(self):
        return self.t.shape

    def bound(self):
        """(part, n), measured on first use: GEMMs that go to the weight-resident kernel never ask."""
        if self.part is None:
            self.part, self.n = _weight_or_measured_max(self.t)
        return self.part, self.n


def is_handle(x):
    """x is an operand handle made by operand() (to be kept for the GEMMs of the backward pass), not a plain tensor."""
    return isinstance(x, (BF16Mat, F32Op, PlanesMat))


def maxabs_partials(x):
    """-> (part, n): n <= 256 partial maxima of |x| for a 2-D fp32 matrix (row-strided views allowed)."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1):
        raise RenetHipError('maxabs_partials needs a 2-D float32 device tensor with unit inner stride')
    rows, cols = x.shape
    n = lib().renet_maxabs_blocks(rows, cols, _ld(x))
    part = torch.empty(n, device=x.device, dtype=torch.float32)
    t0 = _timer.begin() if _timer is not None else None
    _check(lib().renet_maxabs_partials(x.data_ptr(), rows, cols, _ld(x), part.data_ptr(), _stream()),
           'maxabs_partials')
    if t0 is not None:
        _timer.end('maxabs', t0, nbytes=float(rows * cols * 4))
    return part, n


_weight_max = {}               # data_ptr -> (stamp, part, n, guard) of a registered weight


class _StreamGuard(object):
    """Orders consumers on OTHER streams behind the launch that filled a cached device buffer (the first
    bound request after an optimizer step may come from ops._Side's side stream; a main-stream GEMM that then hits
    the cache must not read the partial maxima before that launch has run).  Created right behind the producing
    launch on the producing stream; wait() is free on that same stream."""
    __slots__ = ('stream', 'event')

    def __init__(self, device):
        self.stream = torch.cuda.current_stream(device)
        self.event = torch.cuda.Event()
        self.event.record(self.stream)

    def wait(self, device):
        cur = torch.cuda.current_stream(device)
        if cur != self.stream:
            cur.wait_event(self.event)


_weight_jobs = {'key': None, 'table': None, 'parts': {}}     # one multi-array launch for all registered weights
_weight_objs = {}              # data_ptr -> weakref of the registered weight (its torch version counter is read)


def _measure_all_weights(device):
    """-> {data_ptr: ((epoch, version), part, n)} for EVERY registered weight that is alive, contiguous, 16-byte aligned
    and a multiple of 4 elements long, from one launch (renet_maxabs_partials_multi)."""
    live = {p: r() for p, r in _weight_objs.items()}
    ptrs = sorted(p for p, t in live.items() if t is not None and t.is_cuda and t.device == device and p in _weight_ptrs
                  and t.numel() % 4 == 0 and p % 16 == 0 and t.data_ptr() == p)
    key = (tuple(ptrs), str(device))
    if _weight_jobs['key'] != key:
        rows, parts = [], {}
        for p in ptrs:
            shp = _weight_ptrs[p]
            nb = lib().renet_maxabs_blocks(shp[0], shp[1], shp[1])
            part = torch.empty(nb, device=device, dtype=torch.float32)
            parts[p] = (part, nb)
            rows.append([p, shp[0] * shp[1] // 4, part.data_ptr(), nb])        # {x, n4, part, nblocks | pad}
        _weight_jobs.update(key=key, parts=parts,
                            table=torch.tensor(rows, dtype=torch.int64).to(device) if rows else None)
    if _weight_jobs['table'] is None:
        return {}
    t0 = _timer.begin() if _timer is not None else None
    _check(lib().renet_maxabs_partials_multi(_weight_jobs['table'].data_ptr(), len(ptrs), _stream()),
           'maxabs_partials_multi')
    if t0 is not None:
        _timer.end('maxabs', t0, nbytes=float(sum(_weight_ptrs[p][0] * _weight_ptrs[p][1] * 4 for p in ptrs)))
    guard = _StreamGuard(device)
    return {p: ((_weight_epoch[0], live[p]._version), part, nb, guard) for p, (part, nb) in _weight_jobs['parts'].items()}


def prefetch_weight_bounds(device):
    """Measure every registered weight NOW, on the current stream (parallel.HipAdam.step calls this right behind the
    update, on the stream the step runs on): the f16x3 GEMMs of the next step then only ever hit the cache."""
    if current_mode() == 'f16x3' and _weight_ptrs:
        _weight_max.update(_measure_all_weights(device))


def _weight_or_measured_max(x):
    """A registered weight's maxima are cached until the next optimizer step (a leading row / column block of the
    weight is bounded by the whole weight's); anything else is measured."""
    ptr = x.data_ptr()
    shp = _weight_ptrs.get(ptr)
    if shp is not None and x.dim() == 2 and x.stride(1) == 1 and x.stride(0) == shp[1] and \
            x.shape[0] <= shp[0] and x.shape[1] <= shp[1]:
        stamp = (_weight_epoch[0], x._version)      # in-place torch writes (load_state_dict) bump _version
        ent = _weight_max.get(ptr)
        if ent is None or ent[0][0] != _weight_epoch[0]:
            # first request after an optimizer step: every registered weight changed -- ONE launch measures them all
            _weight_max.update(_measure_all_weights(x.device))
            ent = _weight_max.get(ptr)
        if ent is None or ent[0] != stamp:          # not coverable by the joint launch, or modified since: on its own
            part, n = maxabs_partials(torch.as_strided(x, shp, (shp[1], 1)))
            ent = (stamp, part, n, _StreamGuard(x.device))
            _weight_max[ptr] = ent
        ent[3].wait(x.device)                       # (no-op on the stream that measured)
        return ent[1], ent[2]
    return maxabs_partials(x)


def _f32op(x, bound=None):
    """F32Op of tensor x.  bound: a 1-element device tensor known to bound max |x| (else measured lazily)."""
    if isinstance(x, F32Op):
        return x
    return F32Op(x, bound, 1) if bound is not None else F32Op(x)


_const_bounds = {}


def const_bound(value, device):
    """1-element device tensor holding a known bound (GRU states: 1)."""
    key = (float(value), str(device))
    ent = _const_bounds.get(key)
    if ent is None:
        t = torch.full((1,), float(value), device=device, dtype=torch.float32)
        ent = (t, _StreamGuard(device) if t.is_cuda else None)
        _const_bounds[key] = ent
    if ent[1] is not None:
        ent[1].wait(device)                         # filled on another stream (ops._Side): order this one behind it
    return ent[0]


def operand_like(x, other):
    """Operand handle for x that reuses the magnitude bound of handle `other` (the caller knows |x| <= max |other|
    elementwise: the GRU's dGh against dGi); plain x outside f16x3 mode."""
    if current_mode() != 'f16x3' or not isinstance(other, F32Op):
        return operand(x)
    part, n = other.bound()
    return F32Op(x, part, n)


def _skinny_shape(ta, m, n, k, a, b, split_k):
    """Mirror of renet_gemm_skinny_eligible (gemm_skinny.hip): these shapes run the weight-resident bf16x6 kernel in
    every fp32-class mode and need no magnitude bounds."""
    return (not ta and split_k == 1 and n <= 256 and 16 <= k <= 208 and k % 4 == 0 and m >= 256 and
            a.stride(0) % 4 == 0 and a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 0 and
            os.environ.get('RENET_GEMM_SKINNY', '1') != '0')


# Operand bounds emitted by the kernel that WROTE the tensor (seq_assemble_fwd, concat3_fwd, rgcn_bwd_prep in f16x3
# mode) ride on the tensor OBJECT (`_renet_bound` = (part, n)) and are picked up by operand() instead of a
# renet_maxabs_partials pass; a tensor that arrives without the attribute (re-wrapped by autograd, a view, a copy) is
# simply measured.  The producers' outputs are never modified in place.  RENET_FUSED_BOUNDS=0 turns this off.
def _fused_bounds(t):
    return current_mode() == 'f16x3' and t.is_cuda and t.numel() > 0 and os.environ.get('RENET_FUSED_BOUNDS', '1') != '0'


def _note_bound(t, part, n):
    t._renet_bound = (part, n)


_lazy_shells = {}          # (data_ptr, shape) of an uninitialised fp32 shell -> the BF16Mat it stands for


def lazy_shell(mat, device):
    """fp32 tensor of mat's logical shape whose VALUES ARE NEVER WRITTEN: the autograd-visible stand-in of a tensor
    that exists only as a bf16 operand matrix (bf16-storage mode); operand() resolves it to `mat`."""
    import weakref
    x = torch.empty(mat.R, mat.C, device=device, dtype=torch.float32)
    x._renet_bf16 = mat
    key = (x.data_ptr(), (mat.R, mat.C))
    _lazy_shells[key] = mat
    # the address-keyed entry lives exactly as long as the shell OBJECT (a shell dropped before it reached
    # operand() must not leave an entry that a later fp32 tensor at the same address and shape would resolve to)
    weakref.finalize(x, _drop_lazy_shell, key, mat)
    return x


def _drop_lazy_shell(key, mat):
    if _lazy_shells.get(key) is mat:
        del _lazy_shells[key]


def operand(x, bound=None):
    """GEMM operand for activation tensor x: in bf16-storage mode its bf16 copy (packed ONCE, to be handed to every
    GEMM that consumes x), in f16x3 mode x with the bound on its magnitude (ONE pass over x, or `bound`: a 1-element
    device tensor the producer knows to bound max |x|), otherwise x itself."""
    if isinstance(x, (BF16Mat, F32Op)):
        return x
    lazy = getattr(x, '_renet_bf16', None)          # a producer already wrote the bf16 form (ops.SeqAssembleFn)
    if lazy is None:
        # the attribute rides on the Python object; should autograd ever hand over a re-wrapped tensor, the shell is
        # still recognised by its storage -- an uninitialised shell must never be packed
        lazy = _lazy_shells.pop((x.data_ptr(), tuple(x.shape)), None)
    else:
        _lazy_shells.pop((x.data_ptr(), tuple(x.shape)), None)
    if lazy is not None:
        return lazy
    if current_mode() == 'bf16s':
        return pack_bf16(x)
    if current_mode() == 'f16x3' and x.is_cuda and x.dim() == 2:
        noted = getattr(x, '_renet_bound', None)
        if noted is not None and bound is None:
            return F32Op(x, noted[0], noted[1])
        return _f32op(x, bound)
    return x


def gemm_bf16s(a, b, ta=False, tb=False, out=None, bias=None, alpha=1.0, beta=0.0, split_k=None):
    """gemm() on bf16-STORED operands (fp32 tensors are packed first; registered weights come from the cache).
    A leading block of a cached matrix along the CONTRACTION dimension is only valid when the block ends on a 64-wide
    k stage (or at the matrix edge, where the padding is zero) -- otherwise it is packed afresh."""
    def prep(x, k_axis):
        m, r, c = _as_bf16(x)
        klen, kfull = (r, m.R) if k_axis == 0 else (c, m.C)
        if klen != kfull and (klen % 64) != 0:
            m = pack_bf16(x)
            r, c = m.R, m.C
        return m, r, c
    pa, ar, ac = prep(a, 0 if ta else 1)
    pb, br, bc = prep(b, 1 if tb else 0)
    m, k = (ac, ar) if ta else (ar, ac)
    n, k2 = (br, bc) if tb else (bc, br)
    if k != k2:
        raise RenetHipError('gemm inner dimensions differ: %d vs %d' % (k, k2))
    if out is None:
        if beta != 0.0:
            raise RenetHipError('beta != 0 needs an output tensor')
        out = torch.empty(m, n, device=pa.p.device, dtype=torch.float32)
    if split_k is None:
        split_k = auto_split_k(m, n, k)
    ws_ptr, ws_bytes = None, 0
    if split_k > 1:
        ws_bytes = lib().renet_gemm_workspace(m, n, split_k)
        ws = torch.empty(ws_bytes // 4, device=out.device, dtype=torch.float32)
        ws_ptr = ws.data_ptr()
    t0 = _timer.begin() if _timer is not None else None
    _check(lib().renet_gemm_bf16s(int(ta), int(not tb), m, n, k, float(alpha), pa.p.data_ptr(), pa.p.shape[1],
                                  pb.p.data_ptr(), pb.p.shape[1], float(beta), out.data_ptr(), _ld(out), _f32(bias),
                                  split_k, ws_ptr, ws_bytes, _stream()), 'gemm_bf16s')
    if t0 is not None:
        _timer.end('gemm_f32', t0, flops=2.0 * m * n * k, tag=(int(ta), int(tb), m, n, k, split_k))
    return out


def gemm(a, b, ta=False, tb=False, out=None, bias=None, alpha=1.0, beta=0.0, split_k=None, mode=None):
    """out = alpha * op(a) @ op(b) + bias + beta * out.  a/b may be row-strided views.
    split_k=None picks the deterministic split-K factor automatically."""
    if (mode or current_mode()) == 'bf16s' or isinstance(a, BF16Mat) or isinstance(b, BF16Mat):
        return gemm_bf16s(a, b, ta=ta, tb=tb, out=out, bias=bias, alpha=alpha, beta=beta, split_k=split_k)
    md = mode or current_mode()
    opa, opb = a, b
    if isinstance(a, F32Op):
        a = a.t
    if isinstance(b, F32Op):
        b = b.t
    if not (a.is_cuda and b.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32):
        raise RenetHipError('gemm operands must be float32 device tensors')
    m, k = (a.shape[1], a.shape[0]) if ta else (a.shape[0], a.shape[1])
    k2, n = (b.shape[1], b.shape[0]) if tb else (b.shape[0], b.shape[1])
    if k != k2:
        raise RenetHipError('gemm inner dimensions differ: %d vs %d' % (k, k2))
    if out is None:
        out = torch.empty(m, n, device=a.device, dtype=torch.float32)
        if beta != 0.0:
            raise RenetHipError('beta != 0 needs an output tensor')
    if split_k is None:
        split_k = auto_split_k(m, n, k)
    ws_ptr, ws_bytes = None, 0
    if split_k > 1:
        ws_bytes = lib().renet_gemm_workspace(m, n, split_k)
        ws = torch.empty(ws_bytes // 4, device=a.device, dtype=torch.float32)
        ws_ptr = ws.data_ptr()
    if md == 'f16x3' and _skinny_shape(ta, m, n, k, a, b, split_k):
        md = 'bf16x6'
    if md == 'f16x3':
        (pa, na), (pb, nb) = _f32op(opa).bound(), _f32op(opb).bound()     # (measured outside the GEMM's timer)
    t0 = _timer.begin() if _timer is not None else None
    if md == 'f16x3':
        _check(lib().renet_gemm_f32_h3(int(ta), int(tb), m, n, k, float(alpha), a.data_ptr(), _ld(a), b.data_ptr(),
                                       _ld(b), float(beta), out.data_ptr(), _ld(out), _f32(bias), split_k, ws_ptr,
                                       ws_bytes, pa.data_ptr(), na, pb.data_ptr(), nb, _stream()),
               'gemm_f32_h3')
    else:
        fn = lib().renet_gemm_f32_split if md == 'bf16x6' else lib().renet_gemm_bf16 if md == 'bf16' else \
            lib().renet_gemm_f32
        _check(fn(int(ta), int(tb), m, n, k, float(alpha), a.data_ptr(), _ld(a), b.data_ptr(),
                  _ld(b), float(beta), out.data_ptr(), _ld(out), _f32(bias), split_k, ws_ptr,
                  ws_bytes, _stream()), 'gemm_f32')
    if t0 is not None:
        _timer.end('gemm_f32', t0, flops=2.0 * m * n * k, tag=(int(ta), int(tb), m, n, k, split_k))
    return out


def colsum(x, out=None, beta=0.0):
    """out = beta * out + column sums of x (fp32 tensor or BF16Mat)."""
    if isinstance(x, BF16Mat):
        m, n = x.R, x.C
        if out is None:
            if beta != 0.0:
                raise RenetHipError('beta != 0 needs an output tensor')
            out = torch.empty(n, device=x.p.device, dtype=torch.float32)
        nbytes = lib().renet_colsum_workspace(m, n)
        ws = torch.empty(nbytes // 4, device=x.p.device, dtype=torch.float32) if nbytes else None
        _check(lib().renet_colsum_bf16(x.p.data_ptr(), m, n, x.p.shape[1], _f32(out), float(beta),
                                       ws.data_ptr() if nbytes else None, nbytes, _stream()), 'colsum_bf16')
        return out
    m, n = x.shape
    if out is None:
        if beta != 0.0:
            raise RenetHipError('beta != 0 needs an output tensor')
        out = torch.empty(n, device=x.device, dtype=torch.float32)
    nbytes = lib().renet_colsum_workspace(m, n)
    ws = torch.empty(nbytes // 4, device=x.device, dtype=torch.float32) if nbytes else None
    _check(lib().renet_colsum(x.data_ptr(), m, n, _ld(x), _f32(out), float(beta), ws.data_ptr() if nbytes else None,
                              nbytes, _stream()), 'colsum')
    return out


def scale_by_device_scalar(x, g, bound_in=None):
    """x *= g in place, g a 0-dim / 1-element DEVICE tensor (no host sync; g == 1 leaves x untouched).
    bound_in: a host float bounding max |x| before the scaling -> returns the 1-element device tensor |g| * bound_in
    (fp32 tensors only), the operand bound of the f16x3 GEMMs that consume x; otherwise returns x."""
    if bound_in is not None and not isinstance(x, BF16Mat):
        if not x.is_contiguous():
            raise RenetHipError('scale_by_device_scalar needs a contiguous tensor')
        out = torch.empty(1, device=x.device, dtype=torch.float32)
        _check(lib().renet_scale_by_device_scalar_bound(_f32(x), x.numel(), _f32(g), float(bound_in), out.data_ptr(),
                                                        _stream()), 'scale_by_device_scalar_bound')
        return out
    if isinstance(x, BF16Mat):
        _check(lib().renet_scale_bf16_by_device_scalar(x.p.data_ptr(), x.p.numel(), _f32(g), _stream()),
               'scale_bf16_by_device_scalar')
        return x
    if not x.is_contiguous():
        raise RenetHipError('scale_by_device_scalar needs a contiguous tensor')
    _check(lib().renet_scale_by_device_scalar(_f32(x), x.numel(), _f32(g), _stream()), 'scale_by_device_scalar')
    return x


def seq_assemble_fwd(h2, ent, rel, glob, subj_row, row_ent, row_rel, glob_row, drop_p, seed_x, seed_xr):
    s, d = subj_row.numel(), h2.shape[1]
    x = torch.empty(s, 4 * d, device=h2.device, dtype=torch.float32)
    xr = torch.empty(s, 3 * d, device=h2.device, dtype=torch.float32)
    if _fused_bounds(x):                         # the operand bounds of X / Xr come out of the kernel that writes them
        nparts = lib().renet_bound_parts(s * d)
        parts = torch.empty(2, nparts, device=h2.device, dtype=torch.float32)
        _check(lib().renet_seq_assemble_fwd_bounds(_f32(h2), _f32(ent), _f32(rel), _f32(glob), _i32(subj_row),
                                                   _i32(row_ent), _i32(row_rel), _i32(glob_row), s, d, float(drop_p),
                                                   int(seed_x), int(seed_xr), _f32(x), _f32(xr), parts[0].data_ptr(),
                                                   parts[1].data_ptr(), _stream()), 'seq_assemble_fwd_bounds')
        _note_bound(x, parts[0], nparts)
        _note_bound(xr, parts[1], nparts)
        return x, xr
    _check(lib().renet_seq_assemble_fwd(_f32(h2), _f32(ent), _f32(rel), _f32(glob), _i32(subj_row),
                                        _i32(row_ent), _i32(row_rel), _i32(glob_row), s, d, float(drop_p),
                                        int(seed_x), int(seed_xr), _f32(x), _f32(xr), _stream()),
           'seq_assemble_fwd')
    return x, xr


def seq_assemble_fwd_bf16(h2, ent, rel, glob, subj_row, row_ent, row_rel, glob_row, drop_p, seed_x, seed_xr):
    """seq_assemble_fwd with X / Xr written as bf16 operand matrices (-> BF16Mat, BF16Mat); no fp32 copy exists."""
    s, d = subj_row.numel(), h2.shape[1]
    x, xr = bf16_empty(s, 4 * d, h2.device), bf16_empty(s, 3 * d, h2.device)
    _check(lib().renet_seq_assemble_fwd_bf16(_f32(h2), _f32(ent), _f32(rel), _f32(glob), _i32(subj_row), _i32(row_ent),
                                             _i32(row_rel), _i32(glob_row), s, d, float(drop_p), int(seed_x),
                                             int(seed_xr), x.p.data_ptr(), x.p.shape[1], xr.p.data_ptr(),
                                             xr.p.shape[1], x.p.shape[0], _stream()), 'seq_assemble_fwd_bf16')
    return x, xr


def softmax_ce_bf16(logits, target, grad_scale, row_loss=None):
    """-> (row_loss[B], BF16Mat (softmax - onehot) * grad_scale); the fp32 logits are not modified."""
    b, c = logits.shape
    if row_loss is None:
        row_loss = torch.empty(b, device=logits.device, dtype=torch.float32)
    dl = bf16_empty(b, c, logits.device)
    _check(lib().renet_softmax_ce_bf16(logits.data_ptr(), _i32(target), b, c, _ld(logits), float(grad_scale),
                                       _f32(row_loss), dl.p.data_ptr(), dl.p.shape[1], min((b + 63) & ~63, dl.p.shape[0]),
                                       _stream()), 'softmax_ce_bf16')
    return row_loss, dl


def seq_assemble_bwd(dx, dxr, step_off, num_steps, num_seq, d, drop_p, seed_x, seed_xr):
    """-> (dRows[S,D], dEntSeq[num_seq,D], dRelSeq[num_seq,D]); step_off: device int32 [L+1]."""
    s = dx.shape[0]
    d_rows = torch.empty(s, d, device=dx.device, dtype=torch.float32)
    d_ent = torch.empty(num_seq, d, device=dx.device, dtype=torch.float32)
    d_rel = torch.empty(num_seq, d, device=dx.device, dtype=torch.float32)
    _check(lib().renet_seq_assemble_bwd(_f32(dx), _f32(dxr), _i32(step_off), num_steps, s, num_seq, d,
                                        float(drop_p), int(seed_x), int(seed_xr), _f32(d_rows), _f32(d_ent),
                                        _f32(d_rel), _stream()), 'seq_assemble_bwd')
    return d_rows, d_ent, d_rel


def gru_fwd(gi, step_off_host, hdim, w_hh, b_hh, out_rows=0):
    """gi [S,3H] packed; step_off_host: ctypes int32 array (L+1).  Returns (h_last[max(B, out_rows), H] with
    rows >= B zero, saved[S,5H]).  (One-problem form of gru_fwd_layouts, which also selects the bf16-mode kernels.)"""
    hs, svs = gru_fwd_layouts([gi], [step_off_host], hdim, [w_hh], [b_hh], [out_rows])
    return hs[0], svs[0]


def gru_bwd(dh_last, step_off_host, hdim, w_hh, saved):
    d_gis, d_ghs = gru_bwd_layouts([dh_last], [step_off_host], hdim, [w_hh], [saved])
    return d_gis[0], d_ghs[0]


def _ptrs(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def gru_fwd_multi(gis, step_off_host, hdim, w_hhs, b_hhs, out_rows=0):
    """n GRUs over the same packed layout in one launch -> ([h_last...], [saved...]); h_last has
    max(B, out_rows) rows, the ones past B zero."""
    L = len(step_off_host) - 1
    b = step_off_host[1] - step_off_host[0] if L > 0 else 0
    out_rows = max(int(out_rows), b)
    dev = gis[0].device
    hs = [torch.empty(out_rows, hdim, device=dev, dtype=torch.float32) for _ in gis]
    svs = [torch.empty(g.shape[0], 5 * hdim, device=dev, dtype=torch.float32) for g in gis]
    for t in list(gis) + list(w_hhs) + list(b_hhs):
        _f32(t)
    nbytes = len(gis) * lib().renet_gru_workspace(b, hdim)
    ws = torch.empty(max(nbytes // 4, 1), device=dev, dtype=torch.float32)
    t0 = _timer.begin() if _timer is not None else None
    _check(lib().renet_gru_fwd_multi(len(gis), _ptrs(gis), ctypes.cast(step_off_host, c_void_p), L, hdim,
                                     _ptrs(w_hhs), _ptrs(b_hhs), _ptrs(hs), out_rows, _ptrs(svs), ws.data_ptr(),
                                     nbytes, _stream()), 'gru_fwd_multi')
    if t0 is not None:      # recurrence only: h_{t-1} @ W_hh^T per step, 2 * 3H * H flops per packed row
        _timer.end('gru_recurrence', t0, flops=sum(2.0 * 3 * hdim * hdim * g.shape[0] for g in gis))
    return hs, svs


def gru_bwd_multi(dh_lasts, step_off_host, hdim, w_hhs, saveds):
    L = len(step_off_host) - 1
    n = len(dh_lasts)
    dev = saveds[0].device
    d_gis = [torch.empty(s.shape[0], 3 * hdim, device=dev, dtype=torch.float32) for s in saveds]
    d_ghs = [torch.empty(s.shape[0], 3 * hdim, device=dev, dtype=torch.float32) for s in saveds]
    for t in list(dh_lasts) + list(w_hhs) + list(saveds):
        _f32(t)
    nbytes = n * lib().renet_gru_workspace(dh_lasts[0].shape[0], hdim)
    ws = torch.empty(max(nbytes // 4, 1), device=dev, dtype=torch.float32)
    t0 = _timer.begin() if _timer is not None else None
    _check(lib().renet_gru_bwd_multi(n, _ptrs(dh_lasts), ctypes.cast(step_off_host, c_void_p), L, hdim,
                                     _ptrs(w_hhs), _ptrs(saveds), _ptrs(d_gis), _ptrs(d_ghs), ws.data_ptr(),
                                     nbytes, _stream()), 'gru_bwd_multi')
    if t0 is not None:      # dh_{t-1} += d_gates @ W_hh per step: the same 2 * 3H * H flops per packed row
        _timer.end('gru_recurrence', t0, flops=sum(2.0 * 3 * hdim * hdim * s_.shape[0] for s_ in saveds))
    return d_gis, d_ghs


def _offs(step_offs):
    return ((ctypes.c_void_p * len(step_offs))(*[ctypes.cast(o, c_void_p).value for o in step_offs]),
            (ctypes.c_int * len(step_offs))(*[len(o) - 1 for o in step_offs]))


def gru_fwd_layouts(gis, step_offs, hdim, w_hhs, b_hhs, out_rows):
    """n <= 4 GRUs of up to two packed layouts in one launch (renet_gru_fwd_layouts): step_offs[k] is the host
    offset array of problem k (problems of one layout pass the SAME object), out_rows[k] its h_last height.
    -> ([h_last...], [saved...])."""
    n = len(gis)
    dev = gis[0].device
    rows = [max(int(r), (o[1] - o[0]) if len(o) > 1 else 0) for r, o in zip(out_rows, step_offs)]
    hs = [torch.empty(r, hdim, device=dev, dtype=torch.float32) for r in rows]
    svs = [torch.empty(g.shape[0], 5 * hdim, device=dev, dtype=torch.float32) for g in gis]
    for t in list(gis) + list(w_hhs) + list(b_hhs):
        _f32(t)
    bmax = max([(o[1] - o[0]) if len(o) > 1 else 0 for o in step_offs] + [0])
    nbytes = n * lib().renet_gru_workspace(int(bmax), hdim)
    ws = torch.empty(max(nbytes // 4, 1), device=dev, dtype=torch.float32)
    so, ls = _offs(step_offs)
    t0 = _timer.begin() if _timer is not None else None
    md = current_mode()
    fn = lib().renet_gru_fwd_layouts_bf16 if md == 'bf16s' else lib().renet_gru_fwd_layouts_f32 if md == 'f32' else \
        lib().renet_gru_fwd_layouts
    _check(fn(n, _ptrs(gis), so, ls, hdim, _ptrs(w_hhs), _ptrs(b_hhs), _ptrs(hs),
                                       (ctypes.c_int * n)(*rows), _ptrs(svs), ws.data_ptr(), nbytes, _stream()),
           'gru_fwd_layouts')
    if t0 is not None:
        _timer.end('gru_recurrence', t0, flops=sum(2.0 * 3 * hdim * hdim * g.shape[0] for g in gis))
    return hs, svs


def gru_bwd_layouts(dh_lasts, step_offs, hdim, w_hhs, saveds, out_bf16=False):
    """out_bf16 (bf16-storage mode only): d_gi / d_gh come back as BF16Mat operand matrices instead of fp32 tensors."""
    n = len(dh_lasts)
    dev = saveds[0].device
    if out_bf16:
        if current_mode() != 'bf16s':
            raise RenetHipError('bf16 GRU gradients exist in bf16-storage mode only')
        d_gis = [bf16_empty(s_.shape[0], 3 * hdim, dev) for s_ in saveds]
        d_ghs = [bf16_empty(s_.shape[0], 3 * hdim, dev) for s_ in saveds]
        lds = {m.p.shape[1] for m in d_gis + d_ghs}
        assert len(lds) == 1
        for m in d_gis + d_ghs:
            _check(lib().renet_bf16_zero_padding(m.p.data_ptr(), m.R, m.C, m.p.shape[1], m.p.shape[0], _stream()),
                   'bf16_zero_padding')
        for t in list(dh_lasts) + list(w_hhs) + list(saveds):
            _f32(t)
        bmax = max([(o[1] - o[0]) if len(o) > 1 else 0 for o in step_offs] + [0])
        nbytes = n * lib().renet_gru_workspace(int(bmax), hdim)
        ws = torch.empty(max(nbytes // 4, 1), device=dev, dtype=torch.float32)
        so, ls = _offs(step_offs)
        t0 = _timer.begin() if _timer is not None else None
        _check(lib().renet_gru_bwd_layouts_bf16out(n, _ptrs(dh_lasts), so, ls, hdim, _ptrs(w_hhs), _ptrs(saveds),
                                                   _ptrs([m.p for m in d_gis]), _ptrs([m.p for m in d_ghs]),
                                                   lds.pop(), ws.data_ptr(), nbytes, _stream()), 'gru_bwd_layouts_bf16out')
        if t0 is not None:
            _timer.end('gru_recurrence', t0, flops=sum(2.0 * 3 * hdim * hdim * s_.shape[0] for s_ in saveds))
        return d_gis, d_ghs
    d_gis = [torch.empty(s.shape[0], 3 * hdim, device=dev, dtype=torch.float32) for s in saveds]
    d_ghs = [torch.empty(s.shape[0], 3 * hdim, device=dev, dtype=torch.float32) for s in saveds]
    for t in list(dh_lasts) + list(w_hhs) + list(saveds):
        _f32(t)
    bmax = max([(o[1] - o[0]) if len(o) > 1 else 0 for o in step_offs] + [0])
    nbytes = n * lib().renet_gru_workspace(int(bmax), hdim)
    ws = torch.empty(max(nbytes // 4, 1), device=dev, dtype=torch.float32)
    so, ls = _offs(step_offs)
    t0 = _timer.begin() if _timer is not None else None
    done = False
    if _fused_bounds(d_gis[0]) and bmax > 0:
        # the kernel that writes dGi also emits its operand bound (one max |dGi| per workgroup)
        nparts = lib().renet_gru_bound_parts(int(bmax))
        parts = torch.empty(n, nparts, device=dev, dtype=torch.float32)
        rc = lib().renet_gru_bwd_layouts_bounds(n, _ptrs(dh_lasts), so, ls, hdim, _ptrs(w_hhs), _ptrs(saveds),
                                                _ptrs(d_gis), _ptrs(d_ghs), _ptrs([parts[k] for k in range(n)]),
                                                ws.data_ptr(), nbytes, _stream())
        if rc == 0:
            for k in range(n):
                _note_bound(d_gis[k], parts[k], nparts)
            done = True
        elif rc != -2:                                   # RENET_ERR_UNSUPPORTED: another recurrence is selected
            _check(rc, 'gru_bwd_layouts_bounds')
    if not done:
        md = current_mode()
        fn = lib().renet_gru_bwd_layouts_bf16 if md == 'bf16s' else lib().renet_gru_bwd_layouts_f32 if md == 'f32' else \
            lib().renet_gru_bwd_layouts
        _check(fn(n, _ptrs(dh_lasts), so, ls, hdim, _ptrs(w_hhs), _ptrs(saveds), _ptrs(d_gis),
                  _ptrs(d_ghs), ws.data_ptr(), nbytes, _stream()), 'gru_bwd_layouts')
    if t0 is not None:
        _timer.end('gru_recurrence', t0, flops=sum(2.0 * 3 * hdim * hdim * s_.shape[0] for s_ in saveds))
    return d_gis, d_ghs


def concat3_fwd(a, ia, hmid, c, ic, drop_p, seed):
    b, d = hmid.shape
    parts = 3 if c is not None else 2
    feat = torch.empty(b, parts * d, device=hmid.device, dtype=torch.float32)
    if _fused_bounds(feat):
        nparts = lib().renet_bound_parts(b * parts * (d // 4))
        part = torch.empty(nparts, device=hmid.device, dtype=torch.float32)
        _check(lib().renet_concat3_fwd_bounds(_f32(a), _i32(ia), _f32(hmid), _f32(c), _i32(ic), b, d, float(drop_p),
                                              int(seed), _f32(feat), part.data_ptr(), _stream()), 'concat3_fwd_bounds')
        _note_bound(feat, part, nparts)
        return feat
    _check(lib().renet_concat3_fwd(_f32(a), _i32(ia), _f32(hmid), _f32(c), _i32(ic), b, d, float(drop_p),
                                   int(seed), _f32(feat), _stream()), 'concat3_fwd')
    return feat


def concat3_bwd(dfeat, d, parts, drop_p, seed):
    b = dfeat.shape[0]
    da = torch.empty(b, d, device=dfeat.device, dtype=torch.float32)
    dh = torch.empty(b, d, device=dfeat.device, dtype=torch.float32)
    dc = torch.empty(b, d, device=dfeat.device, dtype=torch.float32) if parts == 3 else None
    _check(lib().renet_concat3_bwd(_f32(dfeat), b, d, parts, float(drop_p), int(seed), _f32(da), _f32(dh),
                                   _f32(dc), _stream()), 'concat3_bwd')
    return da, dh, dc


def dropout(x, drop_p, seed):
    y = torch.empty_like(x)
    _check(lib().renet_dropout(_f32(x), x.numel(), float(drop_p), int(seed), _f32(y), _stream()), 'dropout')
    return y


def softmax_ce(logits, target, grad_scale, want_grad, row_loss=None):
    """Returns row_loss[B] (written into `row_loss` when given); if want_grad, logits is OVERWRITTEN with
    (softmax - onehot) * grad_scale."""
    b, c = logits.shape
    if row_loss is None:
        row_loss = torch.empty(b, device=logits.device, dtype=torch.float32)
    _check(lib().renet_softmax_ce(logits.data_ptr(), _i32(target), b, c, _ld(logits), float(grad_scale),
                                  _f32(row_loss), logits.data_ptr() if want_grad else None, _stream()),
           'softmax_ce')
    return row_loss


def joint_softmax(logits, num_rels, logits_r, prob_e):
    """logits [n * R, N] -> IN PLACE softmax(row) * softmax(logits_r[e])[r] * prob_e[e]  (model.py:205-209)."""
    nr, n_ent = logits.shape
    n = nr // num_rels
    if nr != n * num_rels or tuple(logits_r.shape) != (n, num_rels) or prob_e.numel() != n:
        raise RenetHipError('joint_softmax: shape mismatch')
    _check(lib().renet_joint_softmax(logits.data_ptr(), _ld(logits), n, num_rels, n_ent, logits_r.data_ptr(),
                                     _ld(logits_r), _f32(prob_e), _stream()), 'joint_softmax')
    return logits


def topk_positive(x, k):
    """Top-k of every row of an fp32 matrix [n, M] -> (values [n, k], int64 indices [n, k]).  Exact selection by an
    order-preserving key (any sign; -0.0 == +0.0; NaNs rank below every number).  The k results come back UNSORTED and
    their slot order -- and which of several elements EQUAL to the k-th value is taken -- depends on the arrival order of
    atomics: sort the result where a deterministic order matters (model.py does not depend on it)."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1):
        raise RenetHipError('topk_positive needs a 2-D float32 device tensor with unit inner stride')
    n, m = x.shape
    vals = torch.empty(n, k, device=x.device, dtype=torch.float32)
    idx = torch.empty(n, k, device=x.device, dtype=torch.int64)
    nbytes = lib().renet_topk_workspace(n)
    ws = torch.empty(nbytes // 4 + 1, device=x.device, dtype=torch.int32)
    _check(lib().renet_topk_positive(x.data_ptr(), x.stride(0) if n > 1 else m, n, m, int(k), vals.data_ptr(),
                                     idx.data_ptr(), ws.data_ptr(), nbytes, _stream()), 'topk_positive')
    return vals, idx


def segment_pool_fwd(h, seg_ptr, num_graphs, is_max):
    d = h.shape[1]
    out = torch.empty(num_graphs, d, device=h.device, dtype=torch.float32)
    arg = torch.empty(num_graphs, d, device=h.device, dtype=torch.int32)
    _check(lib().renet_segment_pool_fwd(_f32(h), _i32(seg_ptr), num_graphs, d, int(is_max), _f32(out),
                                        _i32(arg), _stream()), 'segment_pool_fwd')
    return out, arg


def segment_pool_bwd(dout, seg_ptr, arg, num_graphs, is_max, n):
    d = dout.shape[1]
    dh = torch.empty(n, d, device=dout.device, dtype=torch.float32)
    _check(lib().renet_segment_pool_bwd(_f32(dout), _i32(seg_ptr), _i32(arg), num_graphs, d, int(is_max), n,
                                        _f32(dh), _stream()), 'segment_pool_bwd')
    return dh


def sumsq_partials(g, partial, slot0, n_slots):
    """partial[slot0 : slot0 + n_slots) = per-workgroup sums of squares of the flat fp32 region g (current stream)."""
    _check(lib().renet_sumsq_partials(_f32(g), g.numel(), partial.data_ptr() + 4 * slot0, int(n_slots), _stream()),
           'sumsq_partials')


def adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, max_norm, step, zero_grad=True, norm_out=None,
              grad_scale=1.0, presummed=None):
    """Fused clip + Adam + zero_grad on flat fp32 buffers (in place); the gradient is g * grad_scale.
    presummed = (partial, n): ||g||^2 is the sum of partial[0..n) (sumsq_partials per all-reduce bucket), no norm pass."""
    n = p.numel()
    if presummed is not None:
        part, n_part = presummed
        _check(lib().renet_adam_step_presummed(_f32(p), _f32(g), _f32(m), _f32(v), n, float(lr), float(beta1), float(beta2),
                                               float(eps), float(weight_decay), float(max_norm), float(grad_scale),
                                               int(step), int(zero_grad), _f32(part), int(n_part), _f32(norm_out),
                                               _stream()), 'adam_step_presummed')
        return
    nbytes = lib().renet_adam_workspace(n)
    ws = torch.empty(nbytes // 4, device=p.device, dtype=torch.float32)
    _check(lib().renet_adam_step_scaled(_f32(p), _f32(g), _f32(m), _f32(v), n, float(lr), float(beta1), float(beta2),
                                        float(eps), float(weight_decay), float(max_norm), float(grad_scale), int(step),
                                        int(zero_grad), ws.data_ptr(), nbytes, _f32(norm_out), _stream()), 'adam_step')
