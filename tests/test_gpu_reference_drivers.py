"""The reference's UNMODIFIED pretrain.py / train.py / test.py on the MI355X over the real kernels (the CPU counterpart,
tests/test_reference_drivers.py, runs them over the torch emulation of the wrappers in the build container).

The driver files and the YAGO text files are NOT part of this repository: a GPU session ships them as untracked inputs under the
git-ignored `tools/_trace/refrun/` (copied from /root/reference in the build container: `pretrain.py train.py test.py
data/YAGO/{train,valid,test,stat}.txt`); without them the test skips.  A short schedule (2 + 2 epochs) keeps it under two minutes;
the full README schedule is profiles/r05_e_unmodified_drivers_readme_commands.md."""
import os
import re
import shutil
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFRUN = os.path.join(ROOT, 'tools', '_trace', 'refrun')


def _run(work, script, args, timeout, env=None):
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'run_reference_driver.py'), os.path.join(REFRUN, script)] + args,
                       cwd=work, capture_output=True, text=True, timeout=timeout, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, '%s failed:\n%s\n%s' % (script, r.stdout[-2000:], r.stderr[-3000:])
    return r.stdout


def test_unmodified_reference_drivers_train_and_test_on_the_hip_path(tmp_path):
    need = ['pretrain.py', 'train.py', 'test.py'] + ['data/YAGO/%s.txt' % n for n in ('train', 'valid', 'test', 'stat')]
    if not all(os.path.isfile(os.path.join(REFRUN, f)) for f in need):
        pytest.skip('the reference drivers / YAGO text files were not shipped (tools/_trace/refrun/)')
    assert torch.cuda.is_available()
    import renet_hip
    renet_hip.lib()
    work = str(tmp_path)
    os.makedirs(os.path.join(work, 'data', 'YAGO'))
    os.makedirs(os.path.join(work, 'models', 'YAGO'))
    for n in ('train', 'valid', 'test', 'stat'):
        shutil.copy(os.path.join(REFRUN, 'data', 'YAGO', n + '.txt'), os.path.join(work, 'data', 'YAGO'))
    subprocess.run([sys.executable, os.path.join(ROOT, 're-net_amd', 'preprocess.py'), os.path.join(work, 'data', 'YAGO'), '10'],
                   check=True, capture_output=True, timeout=600)
    common = ['-d', 'YAGO', '--gpu', '0', '--dropout', '0.5', '--n-hidden', '200', '--lr', '1e-3', '--max-epochs', '2',
              '--batch-size', '1024']
    out_p = _run(work, 'pretrain.py', common, 600)
    assert len(re.findall(r'Epoch \d+ \| Loss', out_p)) == 2
    out_t = _run(work, 'train.py', common, 900)
    losses = [float(x) for x in re.findall(r'Epoch \d+ \| Loss ([0-9.]+)', out_t)]
    vmrr = [float(x) for x in re.findall(r'valid MRR \(filtered\): ([0-9.]+)', out_t)]
    assert len(losses) == 2 and losses[1] < losses[0] < 20.0, losses            # (observed 12.6 -> 5.9)
    assert len(vmrr) == 2 and vmrr[-1] > 0.3, vmrr                               # validations of epochs 1 and 2 (observed ~0.5)
    for f in ('rgcn.pth', 'rgcn_graph.pth', 'max1rgcn_global.pth', 'max1rgcn_global2.pth'):
        assert os.path.isfile(os.path.join(work, 'models', 'YAGO', f)), f       # written by the drivers themselves
    out_e = _run(work, 'test.py', ['-d', 'YAGO', '--gpu', '0', '--n-hidden', '200'], 900)
    mrr = float(re.search(r'MRR \(filtered\): ([0-9.]+)', out_e).group(1))
    hits = [float(x) for x in re.findall(r'Hits \(filtered\) @ \d+: ([0-9.]+)', out_e)]
    print('unmodified drivers on the HIP path, 2 + 2 epochs: epoch losses %s, valid MRR %s, TEST MRR %.4f hits %s'
          % (losses, vmrr, mrr, hits))
    assert 0.3 < mrr < 0.9 and len(hits) == 3 and hits[0] <= hits[1] <= hits[2]
    # (the same test.py with the look-ahead off prints the identical metrics in 44 s instead of 4.6 s:
    #  profiles/r05_e_unmodified_drivers_readme_commands.md)
