#!/usr/bin/env python
"""A/B probes of the exact-fp32 GEMM kernel (csrc/gemm.hip).

  python tools/gemm_f32_probe.py build                      # here: variants -> tools/_trace/libf32_<name>.so
  python tools/gemm_f32_probe.py run [M,N,K,ta,tb[,split] ...]   # on the GPU box: times every variant on the shapes

Variants: base; nobar (no s_barrier in the k-loop: wrong results, shows what the barriers cost); nostage (no global
loads / LDS stores in the loop: shows what the staging pieces cost).  RENET_GEMM_F32_KT=16 selects the 16-deep stage.
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'tools', '_trace')
VARIANTS = {'base': [], 'nobar': ['-DF32P_NOBAR'], 'nostage': ['-DF32P_NOSTAGE'],
            'nobar_nostage': ['-DF32P_NOBAR', '-DF32P_NOSTAGE']}
STEP_SHAPES = ['2048,23033,600,0,1', '2048,600,23033,0,0,6', '23033,600,2048,1,0', '16000,600,800,0,1',
               '600,800,16000,1,0,14', '16000,600,600,0,0', '23033,200,200,0,0', '4096,4096,4096,0,1']


def build():
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(ROOT, 're-net_amd', 'csrc', 'gemm.hip')
    procs = []
    for name, flags in VARIANTS.items():
        lib = os.path.join(OUT, 'libf32_%s.so' % name)
        cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
               '-I' + os.path.join(ROOT, 'include')] + flags + [src, '-o', lib]
        procs.append(subprocess.Popen(cmd))
    for p in procs:
        assert p.wait() == 0


def run_one(name, shapes):
    import torch
    lib = ctypes.CDLL(os.path.join(OUT, 'libf32_%s.so' % name))
    vp = ctypes.c_void_p
    lib.renet_gemm_f32.argtypes = [ctypes.c_int] * 5 + [ctypes.c_float, vp, ctypes.c_int, vp, ctypes.c_int,
                                                         ctypes.c_float, vp, ctypes.c_int, vp, ctypes.c_int, vp,
                                                         ctypes.c_size_t, vp]
    dev = torch.device('cuda:0')
    for spec in shapes:
        v = [int(x) for x in spec.split(',')]
        m, n, k, ta, tb = v[:5]
        sk = v[5] if len(v) > 5 else 1
        a = torch.randn((k, m) if ta else (m, k), device=dev)
        b = torch.randn((n, k) if tb else (k, n), device=dev)
        out = torch.empty(m, n, device=dev)
        ws = torch.empty(sk * m * n if sk > 1 else 1, device=dev)

        def call():
            rc = lib.renet_gemm_f32(ta, tb, m, n, k, 1.0, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), 0.0,
                                    out.data_ptr(), n, None, sk, ws.data_ptr(), ws.numel() * 4, None)
            assert rc == 0, rc
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 20
        e0.record()
        for _ in range(it):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / it
        err = ''
        if name == 'base' and m * n * k < 3e10:
            ref = (a.t() if ta else a).double() @ (b.t() if tb else b).double()
            err = ' err %.1e' % float((out.double() - ref).abs().max() / ref.abs().max())
        print('%-14s KT=%-2s %-26s %9.1f us %7.1f TF%s' % (name, os.environ.get('RENET_GEMM_F32_KT', '32'), spec, us,
                                                          2.0 * m * n * k / us / 1e6, err), flush=True)


if __name__ == '__main__':
    if sys.argv[1] == 'build':
        build()
    elif sys.argv[1] == 'one':
        run_one(sys.argv[2], sys.argv[3:] or STEP_SHAPES)
    else:
        shapes = sys.argv[2:] or STEP_SHAPES
        for kt in ('32', '16'):
            for deph in ('1', '0'):
                for name in VARIANTS:
                    print('--- dephase', deph, flush=True)
                    subprocess.call([sys.executable, os.path.abspath(__file__), 'one', name] + shapes,
                                    env=dict(os.environ, RENET_GEMM_F32_KT=kt, RENET_GEMM_F32_DEPHASE=deph))
