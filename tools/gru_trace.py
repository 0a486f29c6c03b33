#!/usr/bin/env python
"""Phase timeline of the persistent GRU forward kernel (s_memtime stamps per wave and time step).

  python tools/gru_trace.py build            # here: hipcc -DRENET_GRU_TRACE -> tools/_trace/librenet_gru_trace.so
  python tools/gru_trace.py run [H] [B] [L]  # on the GPU box: two GRUs of B sequences (bench-like length mix)
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'tools', '_trace')
LIB = os.path.join(OUT, os.environ.get('RENET_TRACE_LIB', 'librenet_gru_trace.so'))
BLOCKS, NW, MAXL, SLOTS = 32, 8, 32, 8


def build():
    os.makedirs(OUT, exist_ok=True)
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-DRENET_GRU_TRACE',
           '-I' + os.path.join(ROOT, 'include'), os.path.join(ROOT, 're-net_amd', 'csrc', 'gru.hip'), '-o', LIB] + sys.argv[2:]
    print(' '.join(cmd))
    subprocess.check_call(cmd)


def run():
    import numpy as np
    import torch
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
    L = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    os.environ['RENET_GRU'] = 'persistent'
    lib = ctypes.CDLL(LIB)
    dev = torch.device('cuda:0')
    rng = np.random.RandomState(0)
    lens = np.sort(np.where(rng.rand(B) < 0.55, L, rng.randint(1, L + 1, B)))[::-1]       # ~55 % full-length histories
    bs = (lens[None, :] > np.arange(L)[:, None]).sum(1)
    off = np.concatenate(([0], np.cumsum(bs))).astype(np.int32)
    S = int(off[-1])
    n = 2
    gis = [torch.randn(S, 3 * H, device=dev) for _ in range(n)]
    whh = [torch.randn(3 * H, H, device=dev) * 0.05 for _ in range(n)]
    bhh = [torch.randn(3 * H, device=dev) * 0.1 for _ in range(n)]
    hs = [torch.empty(B, H, device=dev) for _ in range(n)]
    svs = [torch.empty(S, 5 * H, device=dev) for _ in range(n)]
    vp = ctypes.c_void_p
    lib.renet_gru_workspace.restype = ctypes.c_size_t
    lib.renet_gru_workspace.argtypes = [ctypes.c_int, ctypes.c_int]
    lib.renet_gru_fwd_layouts.argtypes = [ctypes.c_int, vp, vp, vp, ctypes.c_int, vp, vp, vp, vp, vp, vp, ctypes.c_size_t, vp]
    lib.renet_gru_trace_set.argtypes = [vp]
    nbytes = n * lib.renet_gru_workspace(B, H)
    ws = torch.empty(nbytes // 4, device=dev)
    ptrs = lambda ts: (vp * len(ts))(*[t.data_ptr() for t in ts])
    offp = off.ctypes.data_as(vp)
    so = (vp * n)(*[offp.value] * n)
    Ls = (ctypes.c_int * n)(*[L] * n)
    rows = (ctypes.c_int * n)(*[B] * n)

    def go():
        rc = lib.renet_gru_fwd_layouts(n, ptrs(gis), so, Ls, H, ptrs(whh), ptrs(bhh), ptrs(hs), rows, ptrs(svs),
                                       ws.data_ptr(), nbytes, None)
        assert rc == 0, rc
    go()
    torch.cuda.synchronize()
    trace = torch.zeros(BLOCKS * NW * MAXL * SLOTS, dtype=torch.int64, device=dev)
    assert lib.renet_gru_trace_set(trace.data_ptr()) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    go()
    e1.record()
    torch.cuda.synchronize()
    print('H %d, 2 x %d sequences, L %d, S %d: launch %.1f us (traced)' % (H, B, L, S, e0.elapsed_time(e1) * 1e3))
    t = trace.cpu().numpy().reshape(BLOCKS, NW, MAXL, SLOTS).astype(np.int64)
    # workgroups 0..31 hold the 512 longest sequences: alive for all L steps
    alive = [b for b in range(BLOCKS) if (t[b, :, L - 1, 0] > 0).all() and (b + 1) * 16 <= int(bs[L - 1])]
    print('workgroups alive for all %d steps among the first %d: %d' % (L, BLOCKS, len(alive)))
    t = t[alive]
    span = (t[:, 0, L - 1, 6] - t[:, 0, 0, 0]).mean()
    print('first stamp to last stamp: %.0f cycles = %.2f of the launch at 2.4 GHz' % (span, span / 2.4e3 / (e0.elapsed_time(e1) * 1e3)))
    st = t[:, :, 1:L - 1, :]
    step_len = (t[:, :, 2:L, 0] - t[:, :, 1:L - 1, 0])
    print('cycles per step (start to start)        %8.0f' % step_len.mean())
    print('  unit-block loops: prefetch + MFMA     %8.0f   (sum over the wave\'s blocks)' % st[..., 1].mean())
    print('  unit-block loops: gate epilogue       %8.0f' % st[..., 2].mean())
    print('  wait at barrier 1                     %8.0f' % (st[..., 4] - st[..., 3]).mean())
    print('  h copy + plane split                  %8.0f' % (st[..., 5] - st[..., 4]).mean())
    print('  wait at barrier 2                     %8.0f' % (st[..., 6] - st[..., 5]).mean())
    per_wave = (st[..., 3] - st[..., 0]).mean(axis=(0, 2))
    print('  busy until barrier 1, per wave        ' + ' '.join('%6.0f' % x for x in per_wave))


if __name__ == '__main__':
    {'build': build, 'run': run}[sys.argv[1]]()
