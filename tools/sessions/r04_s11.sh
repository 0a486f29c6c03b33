#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4s11
mkdir -p $O
B="--steps 100 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 0 --other-steps 0"
for v in 0 1 0 1; do
RENET_HEAD_DW_SIDE=$v timeout 600 python bench.py $B > $O/bench_dw$v.json 2> $O/bench_dw$v.err
python - <<PY
import json
j=json.loads(open('gpurun_out/r4s11/bench_dw$v.json').read().strip().splitlines()[-1])
print('dw_side=$v', round(j['value']), round(j['ms_per_step'],4), j.get('last_loss'))
PY
done
timeout 600 python tools/infer_bench.py ICEWS18 3 200 > $O/stream.txt 2>&1; grep -v amdgpu.ids $O/stream.txt
