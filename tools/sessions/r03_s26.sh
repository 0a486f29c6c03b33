#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s26
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_streams.py tests/test_gpu_parity.py -m gpu -x -q -k "bounds or side_stream or f16x3" > $O/t.log 2>&1; tail -4 $O/t.log
B="--steps 100 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 0"
timeout 600 python bench.py $B > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/s26/bench.json').read().strip().splitlines()[-1])
print(round(j['value']), round(j['ms_per_step'],4), j.get('last_loss'), j['roofline']['achieved'])
k=j['kernels']
print('   ', {n: (round(v['calls_per_step'],1), round(v['avg_us'],1)) for n,v in k.items()})
PY
