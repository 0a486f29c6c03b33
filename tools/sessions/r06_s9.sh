#!/bin/bash
# round 6, session 9: the C-side launch list (csrc/step.cpp): bit-identity tests, then the plain bench with and without it
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6s9
mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_step_plan.py -x -q) > $O/step_tests.log 2>&1; tail -25 $O/step_tests.log
(timeout 300 python bench.py --plain --steps 50 --warmup 5) > $O/bench_plan.log 2>&1; grep -v amdgpu.ids $O/bench_plan.log | tail -1 | cut -c100-240
python - <<'PY'
import json
d = json.load(open('gpurun_out/bench_detail.json'))
print('plan: value %.0f ms %.4f host_enqueue %.3f launches %s' % (d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'], d.get('launches_per_step')))
PY
(RENET_STEP_PLAN=0 timeout 300 python bench.py --plain --steps 50 --warmup 5) > $O/bench_noplan.log 2>&1
python - <<'PY'
import json
d = json.load(open('gpurun_out/bench_detail.json'))
print('no plan: value %.0f ms %.4f host_enqueue %.3f launches %s' % (d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'], d.get('launches_per_step')))
PY
