#!/bin/bash
# round 6, session 10: C launch list tests again, then plan on / off alternately over 200 steps each
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6s10
mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_step_plan.py -x -q) > $O/step_tests.log 2>&1; tail -5 $O/step_tests.log
for RUN in 1; do
for P in 1 0; do
(RENET_STEP_PLAN=$P timeout 300 python bench.py --plain --steps 200 --warmup 5) > $O/bench_plan$P.log 2>&1
python - <<PY
import json
d = json.load(open('gpurun_out/bench_detail.json'))
print('plan=$P: value %.0f ms %.4f host_enqueue %.3f' % (d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step']))
PY
done
done
