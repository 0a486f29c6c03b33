"""TEST INFRASTRUCTURE: runs ONE unmodified reference script (pretrain.py / train.py / test.py /
data/ICEWS18/get_history_graph.py) in this process, either

  ours  over this repository's API mirror (re-net_amd/ first on sys.path, exactly what
        tools/run_reference_driver.py does) with the C-ABI wrappers emulated in torch-CPU
        (tests/cpu_abi_emulation.py: the build container has no GPU), or
  ref   over the reference's own modules under the test-only DGL shim and the `.cuda()` no-op patch.

    python tests/driver_launcher.py {ours|ref} /root/reference/train.py -d SMALL --gpu -1 ...
"""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def main():
    # torch-version drift, not a property of either implementation: the reference pins torch 1.6 (README.md:37) and
    # its drivers call torch.load(path, map_location=...) on checkpoints that hold numpy arrays (the history lists,
    # train.py:189-195); torch >= 2.6 defaults to weights_only=True and refuses them.  The launcher restores the old
    # default for the run instead of editing the drivers.
    os.environ.setdefault('TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD', '1')
    mode, script = sys.argv[1], os.path.abspath(sys.argv[2])
    sys.argv = [script] + sys.argv[3:]
    sys.path.insert(0, ROOT)
    if mode == 'ours':
        # the two switches tools/run_reference_driver.py turns on for the drivers' calling pattern (round 5): train.py's two
        # model() calls as one merged pass, the per-quadruple evaluate_filter calls from one batched evaluation per timestamp
        os.environ.setdefault('RENET_FUSE_DIRECTIONS', '1')
        os.environ.setdefault('RENET_LOOKAHEAD_EVAL', '1')
        sys.path.insert(0, os.path.join(ROOT, 're-net_amd'))
        sys.path.insert(0, HERE)
        import cpu_abi_emulation
        cpu_abi_emulation.install()
        sys.path.remove(HERE)
        runpy.run_path(script, run_name='__main__')
    elif mode == 'ref':
        from oracle import dgl_shim, ref_loader
        dgl_shim.install()
        sys.path.insert(0, ref_loader.REFERENCE_ROOT)
        with ref_loader.cpu_mode():
            runpy.run_path(script, run_name='__main__')
    else:
        raise SystemExit(__doc__)


if __name__ == '__main__':
    main()
