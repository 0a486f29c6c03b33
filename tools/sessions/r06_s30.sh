#!/bin/bash
# r06 s30: hooks only when a reducer wants them: step-plan tests (incl. the one-rank RCCL run), host enqueue time
O=gpurun_out/r6s30; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_step_plan.py -x -q -m gpu > $O/tests.txt 2>&1; tail -2 $O/tests.txt
timeout 600 python bench.py --steps 100 --warmup 5 > $O/bench.log 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6s30/bench.log').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], 'host_enqueue', d.get('host_enqueue_ms_per_step'), 'median', d.get('value_median'))
PY
