#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4s16
mkdir -p $O
timeout 600 python tools/infer_bench.py advance ICEWS18 > $O/advance.txt 2>&1; grep -v amdgpu.ids $O/advance.txt
timeout 900 python -m pytest tests/test_gpu_config.py tests/test_gpu_builder.py -m gpu -x -q -k "inference or builder" > $O/t.log 2>&1; tail -3 $O/t.log
B="--steps 100 --cpu-steps 0 --e2e-steps 5 --f32-steps 0 --enc-steps 0 --other-steps 0"
timeout 600 python bench.py $B > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r4s16/bench.json').read().strip().splitlines()[-1])
print(round(j['value']), round(j['ms_per_step'],4), 'e2e dev', j['e2e_device_builder'], 'device_build_ms', j['device_build_ms'], 'inline', j['e2e_inline'], 'workers', j['e2e_value'], j['e2e_workers'], 'threads', j['e2e_threads8'], 'host_build_ms', j['host_build_ms'])
PY
