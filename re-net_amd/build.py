"""Builds re-net_amd/csrc/librenet_hip.so (gfx950 only) with hipcc.  No fallback: if hipcc is missing
or a source fails to compile this raises.

    python re-net_amd/build.py [--force]
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(CSRC, 'librenet_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')) + glob.glob(os.path.join(CSRC, '*.cpp')))


def stale():
    if not os.path.isfile(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, '*.h')) + \
        [os.path.join(os.path.dirname(HERE), 'include', 'renet_hip.h')]
    return any(os.path.getmtime(d) > t for d in deps)


WALK_SRC = os.path.join(CSRC, 'listwalk.c')
WALK_LIB = os.path.join(HERE, '_renet_listwalk.so')


def build_listwalk(force=False, verbose=True):
    """The host-side list flattener (csrc/listwalk.c: CPython C API, no device code) -> re-net_amd/_renet_listwalk.so.
    Optional: graph.FlatHistory.from_lists keeps its numpy formulation for a box without a C compiler or Python headers --
    a failure here is reported, not raised (the HIP library below is the part without a fallback)."""
    import sysconfig
    if not force and os.path.isfile(WALK_LIB) and os.path.getmtime(WALK_LIB) >= os.path.getmtime(WALK_SRC):
        return WALK_LIB
    inc = sysconfig.get_paths().get('include')
    extra = []
    try:                                   # numpy's headers, when present: ndarrays read through their struct
        import numpy
        extra = ['-DRENET_LISTWALK_NUMPY', '-I' + numpy.get_include()]
    except Exception:
        pass
    tmp = WALK_LIB + '.%d.tmp' % os.getpid()            # built beside the target, then renamed: concurrent builds stay atomic
    cmd = [os.environ.get('CC', 'gcc'), '-O2', '-shared', '-fPIC', '-I' + str(inc)] + extra + [WALK_SRC, '-o', tmp]
    if verbose:
        print(' '.join(cmd), flush=True)
    try:
        subprocess.check_call(cmd)
        os.replace(tmp, WALK_LIB)
    except (OSError, subprocess.CalledProcessError) as e:
        if os.path.exists(tmp):
            os.remove(tmp)
        print('listwalk.c not built (%s): FlatHistory.from_lists uses its numpy formulation' % e, flush=True)
        return None
    return WALK_LIB


def build(force=False, verbose=True):
    build_listwalk(force, verbose)
    if not force and not stale():
        return LIB
    objs = []
    procs = []
    for src in sources():
        obj = os.path.splitext(src)[0] + '.o'
        flags = FLAGS if src.endswith('.hip') else ['-O3', '-std=c++17', '-fPIC', '-Wall']
        cmd = [HIPCC] + flags + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
        objs.append(obj)
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed on %s' % src)
    cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
