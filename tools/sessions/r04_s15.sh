#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=gpurun_out/r4s15
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_builder.py -m gpu -x -q > $O/t.log 2>&1; tail -3 $O/t.log
timeout 600 python tools/builder_bench.py > $O/builder_bench.txt 2>&1; grep -v amdgpu.ids $O/builder_bench.txt | tail -12
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt -- python $R/tools/builder_bench.py > $R/$O/kt.log 2>&1)
DB=$(find $O/kt -name "*results.db" | head -1); python tools/prof_summary.py "$DB" $O/builder_kernel_stats.md && head -16 $O/builder_kernel_stats.md
find $O -name "*.db" -delete
B="--steps 100 --cpu-steps 0 --e2e-steps 5 --f32-steps 0 --enc-steps 0 --other-steps 0"
timeout 600 python bench.py $B > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r4s15/bench.json').read().strip().splitlines()[-1])
print(round(j['value']), round(j['ms_per_step'],4), 'e2e dev', j['e2e_device_builder'], 'device_build_ms', j['device_build_ms'], 'inline', j['e2e_inline'], 'workers', j['e2e_value'], 'threads', j['e2e_threads8'], 'host_build_ms', j['host_build_ms'])
PY
