#!/bin/bash
# round 6, session 16: persistent planes kernel (v6): tests, micro-benchmark persistent on / off, NODMA ablation, plain bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6s16
mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_planes.py tests/test_gpu_step_plan.py -x -q) > $O/tests.log 2>&1; tail -4 $O/tests.log
for P in 1 0; do
(RENET_P6_PERSISTENT=$P timeout 300 python tools/planes_bench.py) > $O/planes_bench_p$P.log 2>&1; echo "== persistent=$P"; grep -v amdgpu.ids $O/planes_bench_p$P.log | cut -c1-150 | tail -9
done
(RENET_HIP_LIB=$PWD/tools/_trace/p6_nodma.so timeout 300 python tools/planes_bench.py --iters 10) > $O/bench_nodma.log 2>&1; echo "== nodma"; grep -v amdgpu.ids $O/bench_nodma.log | cut -c1-150 | head -4
for RUN in 1 2; do
for P in 1 0; do
(RENET_P6_PERSISTENT=$P timeout 300 python bench.py --plain --steps 100 --warmup 5) > $O/b.log 2>&1
python - <<PY
import json
d = json.load(open('gpurun_out/bench_detail.json'))
print('persistent=$P: value %.0f ms %.4f' % (d['value'], d['ms_per_step']))
PY
done
done
