"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Loads the UNMODIFIED reference modules (/root/reference/{utils,RGCN,Aggregator,model,
global_model}.py) on CPU under `oracle/dgl_shim.py`.  Used in THIS container only:

  * to validate the committed restatement `oracle/renet_oracle.py`, and
  * to generate the golden fixtures under tests/golden/ (tools/make_golden.py).

/root/reference does not exist on the GPU box; everything that runs there uses the committed
restatement + fixtures and never calls into this file (`available()` returns False there).

The reference hard-codes `.cuda()` in 34 places (e.g. model.py:80, utils.py:212,
Aggregator.py:144); `cpu_mode()` turns `Tensor.cuda` into the identity and
`torch.cuda.current_device` into `lambda: 0` for the duration of a `with` block, which is
enough for the reference's training forward/backward, `predict*` and `evaluate_filter` to run
unmodified on CPU tensors.
"""
import contextlib
import importlib.util
import os
import sys

import torch

from . import dgl_shim

REFERENCE_ROOT = os.environ.get('RENET_REFERENCE_ROOT', '/root/reference')
_NAMES = ['utils', 'RGCN', 'Aggregator', 'model', 'global_model']
_cache = None


def available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, 'RGCN.py'))


@contextlib.contextmanager
def cpu_mode():
    """Run reference code on CPU: `.cuda()` -> identity, current_device() -> 0."""
    saved_cuda = torch.Tensor.cuda
    saved_cur = torch.cuda.current_device
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.current_device = lambda: 0
    try:
        yield
    finally:
        torch.Tensor.cuda = saved_cuda
        torch.cuda.current_device = saved_cur


def load():
    """Returns a namespace object with attributes utils, RGCN, Aggregator, model, global_model
    (the unmodified reference modules) and `dgl` (the shim)."""
    global _cache
    if _cache is not None:
        return _cache
    if not available():
        raise RuntimeError('reference tree not present at %s' % REFERENCE_ROOT)
    saved = {n: sys.modules.get(n) for n in _NAMES + ['dgl', 'dgl.function']}
    shim = dgl_shim.install()
    mods = {}
    try:
        for n in _NAMES:
            spec = importlib.util.spec_from_file_location(n, os.path.join(REFERENCE_ROOT, n + '.py'))
            m = importlib.util.module_from_spec(spec)
            sys.modules[n] = m           # the reference uses `from utils import *` etc.
            spec.loader.exec_module(m)
            mods[n] = m
    finally:
        # do not leave generically-named reference modules (or the fake dgl) importable by name
        for n, old in saved.items():
            if old is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = old

    class _NS(object):
        pass
    ns = _NS()
    for n, m in mods.items():
        setattr(ns, n, m)
    ns.dgl = shim
    _cache = ns
    return ns
