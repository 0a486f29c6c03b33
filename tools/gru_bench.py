#!/usr/bin/env python
"""Times the persistent GRU launches (forward, backward) of a given build of the library on the bench's shapes.

  python tools/gru_bench.py build NAME [-DFLAG ...]   # here: tools/_trace/gru_NAME.so (gru.hip rebuilt with the flags)
  python tools/gru_bench.py run [LIB ...]             # on the GPU box: us per launch for each library (default: the product's)
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 're-net_amd', 'csrc')
OUT = os.path.join(ROOT, 'tools', '_trace')


def build():
    name, flags = sys.argv[2], sys.argv[3:]
    os.makedirs(OUT, exist_ok=True)
    obj = os.path.join(OUT, 'gru_%s.o' % name)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-function']
                          + flags + ['-c', os.path.join(CSRC, 'gru.hip'), '-o', obj])
    others = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith('.o') and f != 'gru.o']
    lib = os.path.join(OUT, 'gru_%s.so' % name)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib, obj] + others)
    print(lib)


def run():
    import numpy as np
    import torch
    H, B, L = 200, 2048, 10
    libs = sys.argv[2:] or [os.path.join(CSRC, 'librenet_hip.so')]
    dev = torch.device('cuda:0')
    rng = np.random.RandomState(0)
    lens = np.sort(np.where(rng.rand(B) < 0.25, L, rng.randint(1, 7, B)))[::-1]       # bench-like: S ~ 3.7 per sequence
    bs = (lens[None, :] > np.arange(L)[:, None]).sum(1)
    off = np.concatenate(([0], np.cumsum(bs))).astype(np.int32)
    S = int(off[-1])
    n = 2
    torch.manual_seed(0)
    gis = [torch.randn(S, 3 * H, device=dev) for _ in range(n)]
    whh = [torch.randn(3 * H, H, device=dev) * 0.05 for _ in range(n)]
    bhh = [torch.randn(3 * H, device=dev) * 0.1 for _ in range(n)]
    vp = ctypes.c_void_p
    ptrs = lambda ts: (vp * len(ts))(*[t.data_ptr() for t in ts])
    offp = off.ctypes.data_as(vp)
    so = (vp * n)(*[offp.value] * n)
    Ls = (ctypes.c_int * n)(*[L] * n)
    rows = (ctypes.c_int * n)(*[B] * n)
    ref = None
    for path in libs:
        lib = ctypes.CDLL(os.path.abspath(path))
        lib.renet_gru_workspace.restype = ctypes.c_size_t
        lib.renet_gru_workspace.argtypes = [ctypes.c_int, ctypes.c_int]
        lib.renet_gru_fwd_layouts.argtypes = [ctypes.c_int, vp, vp, vp, ctypes.c_int, vp, vp, vp, vp, vp, vp, ctypes.c_size_t, vp]
        nbytes = n * lib.renet_gru_workspace(B, H)
        ws = torch.empty(nbytes // 4, device=dev)
        hs = [torch.zeros(B, H, device=dev) for _ in range(n)]
        svs = [torch.zeros(S, 5 * H, device=dev) for _ in range(n)]

        def fwd():
            rc = lib.renet_gru_fwd_layouts(n, ptrs(gis), so, Ls, H, ptrs(whh), ptrs(bhh), ptrs(hs), rows, ptrs(svs),
                                           ws.data_ptr(), nbytes, None)
            assert rc == 0, rc
        for _ in range(3):
            fwd()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fwd()
        e1.record()
        torch.cuda.synchronize()
        t_f = e0.elapsed_time(e1) * 1e3 / 50
        lib.renet_gru_bwd_layouts.argtypes = [ctypes.c_int, vp, vp, vp, ctypes.c_int, vp, vp, vp, vp, vp, ctypes.c_size_t, vp]
        dhs = [torch.randn(B, H, device=dev, generator=None) * 0 + 0.01 * (k + 1) for k in range(n)]
        dgi = [torch.zeros(S, 3 * H, device=dev) for _ in range(n)]
        dgh = [torch.zeros(S, 3 * H, device=dev) for _ in range(n)]

        def bwd():
            rc = lib.renet_gru_bwd_layouts(n, ptrs(dhs), so, Ls, H, ptrs(whh), ptrs(svs), ptrs(dgi), ptrs(dgh),
                                           ws.data_ptr(), nbytes, None)
            assert rc == 0, rc
        for _ in range(3):
            bwd()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(50):
            bwd()
        e1.record()
        torch.cuda.synchronize()
        t_b = e0.elapsed_time(e1) * 1e3 / 50
        out = torch.cat([h.reshape(-1) for h in hs] + [s.reshape(-1) for s in svs] + [g.reshape(-1) for g in dgi + dgh])
        if ref is None:
            ref = out
        print('%-28s S %d  forward %.1f us, backward %.1f us per call (incl. the plane split / transpose)   max |diff to first| %.2e'
              % (os.path.basename(path), S, t_f, t_b, float((out - ref).abs().max())))


if __name__ == '__main__':
    {'build': build, 'run': run}[sys.argv[1]]()
