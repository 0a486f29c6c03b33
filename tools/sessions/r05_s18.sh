#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5s18
timeout 600 python tools/yago_full_run.py 0.5 3 3 999 1000 1001 1002 1003 1004 1005 1006 > gpurun_out/r5s18/drop8.json 2> gpurun_out/r5s18/drop8.err
python - <<'PY'
import json, numpy as np
j=json.loads(open('gpurun_out/r5s18/drop8.json').read().strip().splitlines()[-1])
el=np.asarray([r['epoch_loss'] for r in j['runs']]); m=np.asarray([r['mrr'] for r in j['runs']])
print('HIP epoch losses per seed:'); print(np.round(el,4))
print('mean', np.round(el.mean(0),4), 'sd', np.round(el.std(0,ddof=1),4)); print('MRR', np.round(m,4), 'mean %.4f sd %.4f'%(m.mean(), m.std(ddof=1)))
PY
