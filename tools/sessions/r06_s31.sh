#!/bin/bash
# r06 s31: four-row softmax-CE planes writer: tests, isolated timing (tools/planes_bench.py), A/B in the step
O=gpurun_out/r6s31; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_planes.py tests/test_gpu_step_plan.py -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
for v in 4 1; do
  if [ $v = 1 ]; then export RENET_SOFTMAX_ROWS=1; else unset RENET_SOFTMAX_ROWS; fi
  timeout 300 python tools/planes_bench.py 2>&1 | grep -i softmax | sed "s/^/rows=$v  /"
done
for rep in 1 2; do
for v in 1 4; do
  if [ $v = 1 ]; then export RENET_SOFTMAX_ROWS=1; else unset RENET_SOFTMAX_ROWS; fi
  timeout 300 python bench.py --plain --steps 200 --warmup 20 > $O/bench_r${v}_$rep.json 2>/dev/null
done; done
unset RENET_SOFTMAX_ROWS
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o kt -- python $R/bench.py --plain --steps 20 --warmup 3 > $R/$O/prof.log 2>&1)
DB=$(find $O/prof -name "*results.db" | head -1); python tools/prof_summary.py "$DB" $O/stats.md 23
grep -E "softmax|total kernel" $O/stats.md | cut -c1-150
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6s31/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
