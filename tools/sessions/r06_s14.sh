#!/bin/bash
# round 6, session 14: step-plan tests (gradient materialisation), reference-driver tests, list-API rate
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6s14
mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_step_plan.py tests/test_gpu_reference_drivers.py -x -q) > $O/tests.log 2>&1; tail -6 $O/tests.log
(timeout 600 python bench.py --cpu-steps 0 --enc-steps 0 --e2e-steps 0 --f32-steps 0 --other-steps 0 --steps 20) > $O/bench.log 2>&1
python - <<PY
import json
d = json.load(open('gpurun_out/bench_detail.json'))
print('value %.0f  value_list_api %s' % (d['value'], d['value_list_api']))
PY
