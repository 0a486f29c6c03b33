#!/usr/bin/env python
"""Runs one of the reference's UNMODIFIED drivers (train.py / test.py / pretrain.py) on top of the
MI355X modules: `re-net_amd/` is put first on sys.path so that the driver's top-level imports
(`import utils`, `from model import RENet`, `from global_model import RENet_global`, train.py:5-8) resolve
to this repository's implementation; runpy does not add the script's own directory.

    cd <workdir with data/<DS>/ and models/>      # pickles made by `python re-net_amd/preprocess.py <dir>`
    python tools/run_reference_driver.py /path/to/RE-Net/train.py -d YAGO --gpu 0 --n-hidden 200
"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    script = os.path.abspath(sys.argv[1])
    # The drivers were written for torch 1.6 (README.md:37): torch.load(path, map_location=...) on checkpoints that
    # hold numpy arrays (the per-entity history lists, train.py:189-195).  torch >= 2.6 defaults to
    # weights_only=True and refuses those; restore the old default for this run instead of editing the drivers.
    os.environ.setdefault('TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD', '1')
    # train.py:136-138 calls model(..., subject=True) and model(..., subject=False) on the same batch and adds the losses:
    # exactly the contract of RENet.fuse_directions (one merged pass, batch graph built on the device); set
    # RENET_FUSE_DIRECTIONS=0 to run the two calls as two passes
    os.environ.setdefault('RENET_FUSE_DIRECTIONS', '1')
    # test.py:104-139 / train.py:160-172 evaluate one quadruple per call and pass the whole stream as `all_triplets`:
    # RENet.lookahead_eval answers them from one batched evaluation per timestamp (RENET_LOOKAHEAD_EVAL=0: per call)
    os.environ.setdefault('RENET_LOOKAHEAD_EVAL', '1')
    sys.path.insert(0, os.path.join(ROOT, 're-net_amd'))
    sys.argv = [script] + sys.argv[2:]
    runpy.run_path(script, run_name='__main__')


if __name__ == '__main__':
    main()
