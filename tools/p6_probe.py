#!/usr/bin/env python
"""Ablation builds of the planes GEMM (csrc/gemm_p6.h): probe libraries with one component of its k-loop removed, built
into tools/_trace/ (git-ignored, shipped to the GPU box) and selected with RENET_HIP_LIB.  `python tools/p6_probe.py build`
here; on the GPU box `RENET_HIP_LIB=tools/_trace/p6_nodma.so python tools/planes_bench.py`."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 're-net_amd', 'csrc')
OUT = os.path.join(ROOT, 'tools', '_trace')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
VARIANTS = {'nodma': ['-DRENET_P6_NODMA'], 'nomfma': ['-DRENET_P6_NOMFMA'],
            'noread': ['-DRENET_P6_NODMA', '-DRENET_P6_NOREAD']}


def build(extra=None):
    os.makedirs(OUT, exist_ok=True)
    variants = dict(VARIANTS)
    if extra:
        variants = {k: v for k, v in variants.items() if k in extra}
    others = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith('.o') and f != 'gemm_split.o']
    procs = []
    for name, flags in variants.items():
        obj = os.path.join(OUT, 'p6_%s.o' % name)
        cmd = [HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-function'] + flags + \
            ['-c', os.path.join(CSRC, 'gemm_split.hip'), '-o', obj]
        procs.append((name, obj, subprocess.Popen(cmd)))
    for name, obj, p in procs:
        if p.wait() != 0:
            raise SystemExit('hipcc failed on variant %s' % name)
        lib = os.path.join(OUT, 'p6_%s.so' % name)
        subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib, obj] + others)
        print(lib)


if __name__ == '__main__':
    build(sys.argv[2:] or None)
