#!/usr/bin/env python
"""Probe of the bf16x6 / f16x3 GEMM kernels (csrc/gemm_split.hip): how much of a launch is the C-store epilogue?

  python tools/gemm_split_probe.py build     # here: base + -DRENET_PROBE_NOSTORE -> tools/_trace/libsplit_<name>.so
  python tools/gemm_split_probe.py run       # on the GPU box
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'tools', '_trace')
VARIANTS = {'base': [], 'noload': ['-DRENET_PROBE_NOLOAD'], 'noldsw': ['-DRENET_PROBE_NOLDSW'], 'nosplit': ['-DRENET_PROBE_NOSPLIT'],
            'none': ['-DRENET_PROBE_NOLOAD', '-DRENET_PROBE_NOLDSW', '-DRENET_PROBE_NOSPLIT']}
# (earlier: 'nostore': ['-DRENET_PROBE_NOSTORE'] -- the C-store epilogue; results of all of them: profiles/r04_e_bf16x6_kloop.md)
# (round 4, tried and removed: residuals by v_dot2c_f32_bf16 -- not bit-identical, not faster; two k-tiles of load lookahead --
# bit-identical, not faster)        # earlier rounds: 'nostore': ['-DRENET_PROBE_NOSTORE']
SHAPES = ['2048,23033,600,0,1', '2048,600,23033,0,0,6', '23033,600,2048,1,0', '16000,600,800,0,1', '600,800,16000,1,0,14',
          '16000,600,600,0,0', '4096,4096,4096,0,1']


def build(only=None):
    os.makedirs(OUT, exist_ok=True)
    src = [os.path.join(ROOT, 're-net_amd', 'csrc', f) for f in ('gemm_split.hip', 'gemm.hip', 'gemm_skinny.hip')]
    procs = []
    for name, flags in VARIANTS.items():
        if only and name not in only:
            continue
        lib = os.path.join(OUT, 'libsplit_%s.so' % name)
        procs.append(subprocess.Popen(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
                                       '-I' + os.path.join(ROOT, 'include')] + flags + src + ['-o', lib]))
    for p in procs:
        assert p.wait() == 0


def run_one(name, shapes):
    import torch
    lib = ctypes.CDLL(os.path.join(OUT, 'libsplit_%s.so' % name))
    vp = ctypes.c_void_p
    lib.renet_gemm_f32_split.argtypes = [ctypes.c_int] * 5 + [ctypes.c_float, vp, ctypes.c_int, vp, ctypes.c_int,
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
            rc = lib.renet_gemm_f32_split(ta, tb, m, n, k, 1.0, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), 0.0,
                                          out.data_ptr(), n, None, sk, ws.data_ptr(), ws.numel() * 4, None)
            assert rc == 0, rc
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print('%-10s %-26s %9.1f us %7.1f TF' % (name, spec, us, 2.0 * m * n * k / us / 1e6), flush=True)


def compare(names):
    """bitwise comparison of the variants' outputs on the same operands (magnitudes from 1e-30 to 1e30, ragged sizes)"""
    import torch
    dev = torch.device('cuda:0')
    vp = ctypes.c_void_p
    libs = {}
    for name in names:
        lib = ctypes.CDLL(os.path.join(OUT, 'libsplit_%s.so' % name))
        lib.renet_gemm_f32_split.argtypes = [ctypes.c_int] * 5 + [ctypes.c_float, vp, ctypes.c_int, vp, ctypes.c_int,
                                                                   ctypes.c_float, vp, ctypes.c_int, vp, ctypes.c_int, vp,
                                                                   ctypes.c_size_t, vp]
        libs[name] = lib
    torch.manual_seed(5)
    for (m, n, k, ta, tb, sk, mag) in [(300, 700, 333, 0, 1, 1, 1.0), (1024, 1500, 600, 0, 0, 1, 1e-30), (777, 640, 2049, 1, 0, 4, 1e30),
                                      (2048, 23033, 600, 0, 1, 1, 1e-3), (130, 130, 70, 1, 1, 1, 1e-37), (256, 128, 4096, 0, 1, 1, 1e-42)]:
        a = torch.randn((k, m) if ta else (m, k), device=dev) * mag * torch.exp2(torch.randint(-20, 20, (1,), device=dev).float())
        b = torch.randn((n, k) if tb else (k, n), device=dev)
        outs = {}
        for name, lib in libs.items():
            out = torch.empty(m, n, device=dev)
            ws = torch.empty(sk * m * n if sk > 1 else 1, device=dev)
            rc = lib.renet_gemm_f32_split(ta, tb, m, n, k, 1.0, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), 0.0,
                                          out.data_ptr(), n, None, sk, ws.data_ptr(), ws.numel() * 4, None)
            assert rc == 0, rc
            torch.cuda.synchronize()
            outs[name] = out
        base = outs[names[0]]
        for name in names[1:]:
            same = torch.equal(base.view(torch.int32), outs[name].view(torch.int32))
            nd = int((base.view(torch.int32) != outs[name].view(torch.int32)).sum())
            print('compare %s vs %s  %dx%dx%d ta%d tb%d split %d mag %.0e: %s (%d words differ, finite %s)' % (
                names[0], name, m, n, k, ta, tb, sk, mag, 'IDENTICAL' if same else 'DIFFERENT', nd,
                bool(torch.isfinite(base).all())), flush=True)


if __name__ == '__main__':
    if sys.argv[1] == 'build':
        build(sys.argv[2:])
    elif sys.argv[1] == 'cmp':
        compare(list(VARIANTS))
    elif sys.argv[1] == 'one':
        run_one(sys.argv[2], sys.argv[3:] or SHAPES)
    else:
        for name in VARIANTS:
            subprocess.call([sys.executable, os.path.abspath(__file__), 'one', name] + (sys.argv[2:] or SHAPES))
