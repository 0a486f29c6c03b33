#!/bin/bash
# round 5, session 7: the README schedule (pretrain 20 epochs lr 1e-3; train 20 epochs) in the PRODUCT loop over all of YAGO, five
# seeds, validation + test split: the seed spread of the full-length filtered test MRR (context for README.md:169's single run)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s7
mkdir -p $O
RENET_FULL_TEST=1 RENET_FULL_PRE_LR=1e-3 timeout 900 python tools/yago_full_run.py 0.5 20 20 999 1000 1001 1002 1003 > $O/full20.json 2> $O/full20.err
grep -v amdgpu.ids $O/full20.err | grep "seed [0-9]*:" | cut -c1-400; python - <<'PY'
import json, numpy as np
j = json.loads(open('gpurun_out/r5s7/full20.json').read().strip().splitlines()[-1])
t = np.asarray([r['test_mrr'] for r in j['runs']]); v = np.asarray([r['mrr'] for r in j['runs']])
h = np.asarray([r['test_hits'] for r in j['runs']])
print('test MRR', np.round(t, 4), 'mean %.4f sd %.4f' % (t.mean(), t.std(ddof=1)), '| hits@1/3/10 mean', np.round(h.mean(0), 4), '| valid MRR', np.round(v, 4), '| %.0f s' % j['seconds'])
PY
