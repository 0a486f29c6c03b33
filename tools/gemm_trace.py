#!/usr/bin/env python
"""Phase timeline of the bf16x6 GEMM kernels (s_memtime stamps per wave and k-step).

  python tools/gemm_trace.py build                 # here: hipcc -DRENET_GEMM_TRACE -> tools/_trace/librenet_trace.so
  python tools/gemm_trace.py run M N K ta tb        # on the GPU box (two-phase kernel)

Prints, for the traced workgroups, the mean duration (shader cycles) of every phase of the k-loop.
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'tools', '_trace')
LIB = os.path.join(OUT, os.environ.get('RENET_TRACE_LIB', 'librenet_trace.so'))
BLOCKS, STEPS = 64, 320


def build():
    os.makedirs(OUT, exist_ok=True)
    src = [os.path.join(ROOT, 're-net_amd', 'csrc', f) for f in ('gemm_split.hip', 'gemm.hip', 'gemm_skinny.hip')]
    extra = sys.argv[2:]
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
           '-DRENET_GEMM_TRACE', '-I' + os.path.join(ROOT, 'include')] + src + ['-o', LIB] + extra
    print(' '.join(cmd))
    subprocess.check_call(cmd)


def run():
    import numpy as np
    import torch
    m, n, k, ta, tb = [int(x) for x in sys.argv[2:7]]
    os.environ['RENET_GEMM_KERNEL'] = sys.argv[7] if len(sys.argv) > 7 else 'split'
    nw = 4
    lib = ctypes.CDLL(LIB)
    dev = torch.device('cuda:0')
    a = torch.randn((k, m) if ta else (m, k), device=dev)
    b = torch.randn((n, k) if tb else (k, n), device=dev)
    out = torch.empty(m, n, device=dev)
    trace = torch.zeros(BLOCKS * 8 * STEPS * 4, dtype=torch.int64, device=dev)
    vp = ctypes.c_void_p
    lib.renet_gemm_f32_split.argtypes = [ctypes.c_int] * 5 + [ctypes.c_float, vp, ctypes.c_int, vp, ctypes.c_int,
                                                               ctypes.c_float, vp, ctypes.c_int, vp, ctypes.c_int, vp,
                                                               ctypes.c_size_t, vp]
    lib.renet_gemm_trace_set.argtypes = [vp]

    h3 = os.environ['RENET_GEMM_KERNEL'] == 'h3'
    if h3:
        nw = 8 if os.environ.get('RENET_H3_TALL') == '1' else 4
        pa, pb = torch.zeros(256, device=dev), torch.zeros(256, device=dev)
        lib.renet_maxabs_partials.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]
        lib.renet_maxabs_blocks.argtypes = [ctypes.c_int] * 3
        na = lib.renet_maxabs_blocks(a.shape[0], a.shape[1], a.stride(0))
        nb = lib.renet_maxabs_blocks(b.shape[0], b.shape[1], b.stride(0))
        assert lib.renet_maxabs_partials(a.data_ptr(), a.shape[0], a.shape[1], a.stride(0), pa.data_ptr(), None) == 0
        assert lib.renet_maxabs_partials(b.data_ptr(), b.shape[0], b.shape[1], b.stride(0), pb.data_ptr(), None) == 0
        lib.renet_gemm_f32_h3.argtypes = [ctypes.c_int] * 5 + [ctypes.c_float, vp, ctypes.c_int, vp, ctypes.c_int,
                                                                ctypes.c_float, vp, ctypes.c_int, vp, ctypes.c_int, vp,
                                                                ctypes.c_size_t, vp, ctypes.c_int, vp, ctypes.c_int, vp]

    def go():
        if h3:
            rc = lib.renet_gemm_f32_h3(ta, tb, m, n, k, 1.0, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), 0.0,
                                       out.data_ptr(), n, None, 1, None, 0, pa.data_ptr(), na, pb.data_ptr(), nb, None)
        else:
            rc = lib.renet_gemm_f32_split(ta, tb, m, n, k, 1.0, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0),
                                          0.0, out.data_ptr(), n, None, 1, None, 0, None)
        assert rc == 0, rc
    go()
    torch.cuda.synchronize()
    assert lib.renet_gemm_trace_set(trace.data_ptr()) == 0
    trace.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    go()
    e1.record()
    torch.cuda.synchronize()
    print('kernel %.1f us (traced)' % (e0.elapsed_time(e1) * 1e3))
    ref = (a.t() if ta else a).double() @ (b.t() if tb else b).double() if m * n * k < 3e10 else None
    if ref is not None:
        print('err', float((out.double() - ref).abs().max() / ref.abs().max()))
    t = trace.cpu().numpy().reshape(BLOCKS, 8, STEPS, 4).astype(np.int64)
    nkt = (k + 31) // 32
    if True:
        live = t[:, 0, 5, 0] != 0
        print('traced workgroups:', int(live.sum()))
        t = t[live]
        st = t[:, :nw, :nkt, :]
        s0, s1, s2, s3 = st[..., 0], st[..., 1], st[..., 2], st[..., 3]
        nxt = np.concatenate([s0[:, :, 1:], s0[:, :, -1:]], axis=2)
        sl = (slice(None), slice(None), slice(4, nkt - 4))
        print('convert  work %7.0f cyc   barrier wait %7.0f' % ((s1 - s0)[sl].mean(), (s2 - s1)[sl].mean()))
        print('mfma     work %7.0f cyc   barrier wait %7.0f' % ((s3 - s2)[sl].mean(), (nxt - s3)[sl].mean()))
        print('mean k-tile %7.0f cyc' % ((s0[:, :, nkt - 5] - s0[:, :, 5]) / float(nkt - 10)).mean())
        if h3 and os.environ.get('RENET_FINE'):
            f = t[:, :nw, 300, :]
            print('MFMA phase of the last k-tile: g0 -> g7 %6.0f   g7 -> g11 %6.0f   g11 -> g23 %6.0f cycles' % (
                (f[..., 1] - f[..., 0]).mean(), (f[..., 2] - f[..., 1]).mean(), (f[..., 3] - f[..., 2]).mean()))
        if h3:
            for w in range(nw):
                print('  wave %d: store %6.0f  wait %6.0f  mfma %6.0f  wait %6.0f' % (
                    w, (s1 - s0)[:, w, 4:nkt - 4].mean(), (s2 - s1)[:, w, 4:nkt - 4].mean(),
                    (s3 - s2)[:, w, 4:nkt - 4].mean(), (nxt - s3)[:, w, 4:nkt - 4].mean()))
            return
        # co-residence: which traced workgroups share a CU (same HW_ID cu/se/xcc bits), and their phase offset
        hw = t[:, 0, STEPS - 1, 0]
        xcc = t[:, 0, STEPS - 1, 1]
        key = [(int(x), (int(h) >> 8) & 0xf, (int(h) >> 13) & 0x7, (int(h) >> 16) & 0x3) for h, x in zip(hw, xcc)]
        groups = {}
        for i, kk in enumerate(key):
            groups.setdefault(kk, []).append(i)
        shown = 0
        for kk, ids in groups.items():
            if len(ids) >= 2 and shown < 6:
                i, j = ids[0], ids[1]
                d = (s2[j, 0, 8:nkt - 8] - s2[i, 0, 8:nkt - 8])
                per = float((s0[i, 0, nkt - 5] - s0[i, 0, 5]) / float(nkt - 10))
                print('CU %s: blocks %d,%d  mfma-start offset mean %8.0f (std %6.0f) of period %6.0f' % (
                    str(kk), i, j, d.mean(), d.std(), per))
                shown += 1


if __name__ == '__main__':
    build() if sys.argv[1] == 'build' else run()
