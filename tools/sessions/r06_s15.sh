#!/bin/bash
# round 6, session 15: the reference-API loop (value_list_api) with the C launch list on / off, same box
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6s15
mkdir -p $O
for RUN in 1 2; do
for P in 1 0; do
(RENET_STEP_PLAN=$P timeout 600 python bench.py --cpu-steps 0 --enc-steps 0 --e2e-steps 0 --f32-steps 0 --other-steps 0 --steps 20 --list-steps 40) > $O/bench$P.log 2>&1
python - <<PY
import json
d = json.load(open('gpurun_out/bench_detail.json'))
print('plan=$P value %.0f  list api %.0f triples/s %.3f ms' % (d['value'], d['value_list_api']['value'], d['value_list_api']['ms_per_step']))
PY
done
done
