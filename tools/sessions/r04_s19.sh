#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4s19
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_streams.py tests/test_gpu_gather.py -m gpu -x -q > $O/t.log 2>&1; tail -5 $O/t.log
B="--steps 100 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 0 --other-steps 0"
for v in 1 0 1 0; do
RENET_DEFER_GRADS=$v timeout 600 python bench.py $B > $O/bench_d$v.json 2> $O/bench_d$v.err
python - <<PY
import json
j=json.loads(open('gpurun_out/r4s19/bench_d$v.json').read().strip().splitlines()[-1])
print('defer=$v', round(j['value']), round(j['ms_per_step'],4), j.get('last_loss'))
PY
done
