#!/bin/bash
# round 6, session 12: interleaved A/B of the step schedules: autograd path, C list (early fork), C list (late fork)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6s12
mkdir -p $O
for RUN in 1 2 3; do
for CFG in "0 0" "1 0" "1 1"; do
set -- $CFG
(RENET_STEP_PLAN=$1 RENET_STEP_LATE_FORK=$2 timeout 300 python bench.py --plain --steps 100 --warmup 5) > $O/b.log 2>&1
python - <<PY
import json
d = json.load(open('gpurun_out/bench_detail.json'))
print('run $RUN plan=$1 late=$2: value %.0f ms %.4f' % (d['value'], d['ms_per_step']))
PY
done
done
