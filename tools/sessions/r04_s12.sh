#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4s12
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gather.py tests/test_gpu_parity.py -m gpu -x -q -k "gather or rgcn or training_step or adjoint" > $O/t.log 2>&1; tail -4 $O/t.log
(timeout 600 python tools/gather_bench.py both --json $O/gather_both.json) > $O/gather_both.log 2>&1; grep -v "JSON\|amdgpu.ids" $O/gather_both.log | tail -12
(timeout 600 python tools/gather_bench.py global --json $O/gather_global.json) > $O/gather_global.log 2>&1; grep -v "JSON\|amdgpu.ids" $O/gather_global.log | tail -8
B="--steps 100 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 0 --other-steps 0"
timeout 600 python bench.py $B > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r4s12/bench.json').read().strip().splitlines()[-1])
print(round(j['value']), round(j['ms_per_step'],4), j.get('last_loss'))
for k,v in j['roofline_rgcn_gather'].items(): print(k, round(v['avg_us'],1), 'strict %.3f' % v['frac_strict'])
PY
