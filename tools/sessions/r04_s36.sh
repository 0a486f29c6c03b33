#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4s36
mkdir -p $O
B="python bench.py --steps 100 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 0 --other-steps 0"
timeout 600 $B > $O/bench.json 2> $O/bench.err
python - <<PY
import json
j=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print(round(j['value']), round(j['ms_per_step'],3), 'host enqueue ms/step', j['host_enqueue_ms_per_step'])
PY
