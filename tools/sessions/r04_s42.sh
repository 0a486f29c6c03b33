#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4s42
mkdir -p $O
timeout 100 python bench.py --steps 100 --cpu-steps 2 --e2e-steps 0 --f32-steps 0 --enc-steps 0 --other-steps 0 > $O/bench.json 2> $O/bench.err
python - <<PY
import json
j=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print(round(j['value']), round(j['ms_per_step'],3), 'frac %.3f' % j['roofline']['frac'], 'traffic', j['roofline']['traffic'], 'parity', j['parity']['rel_err'], j['parity']['grad_rel_err'])
PY
