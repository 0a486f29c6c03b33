#!/bin/bash
# round 6, session 13: one-rank RCCL group (RENET_FORCE_REDUCER=1: the reducer's collectives really go through RCCL), C launch
# list on / off; then the default bench command's new blocks (value_median, value_list_api, train-mode cpu baseline)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6s13
mkdir -p $O
for RUN in 1; do
for P in 1 0; do
(RENET_FORCE_REDUCER=1 RENET_STEP_PLAN=$P timeout 300 python bench.py --plain --steps 60 --warmup 5) > $O/rccl_plan$P.log 2>&1
python - <<PY
import json
d = json.load(open('gpurun_out/bench_detail.json'))
print('one-rank RCCL, plan=$P: value %.0f ms %.4f median %.4f host_enqueue %.3f' % (d['value'], d['ms_per_step'], d['value_median']['ms_per_step'], d['host_enqueue_ms_per_step']))
PY
done
done
(timeout 1200 python bench.py --f32-steps 0 --other-steps 0) > $O/bench.log 2>&1; grep -v amdgpu.ids $O/bench.log | tail -1 | cut -c1-3000
cp gpurun_out/bench_detail.json $O/bench_detail.json
