#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4s22
mkdir -p $O
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else None)"
B="--steps 60 --cpu-steps 0 --e2e-steps 5 --f32-steps 0 --enc-steps 0 --other-steps 0"
for v in 1 0 1 0; do
RENET_BUILDER_PRIO=$v timeout 600 python bench.py $B > $O/bench_p$v.json 2> $O/bench_p$v.err; tail -c 200 $O/bench_p$v.err
python - <<PY
import json
j=json.loads(open('gpurun_out/r4s22/bench_p$v.json').read().strip().splitlines()[-1])
print('builder_prio=$v', round(j['value']), 'e2e dev', round(j['e2e_device_builder']), round(j['e2e_device_builder']/j['value'],3))
PY
done
