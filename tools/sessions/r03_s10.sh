#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s10
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_builder.py -m gpu -x -q > $O/t_builder.log 2>&1; tail -3 $O/t_builder.log
timeout 900 python bench.py --steps 60 --cpu-steps 0 --f32-steps 0 --enc-steps 0 --e2e-steps 10 > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/s10/bench.json').read().strip().splitlines()[-1])
print({k:j.get(k) for k in ('value','ms_per_step','host_build_ms','e2e_inline','e2e_value','e2e_threads8','e2e_device_builder','device_build_ms')})
PY
