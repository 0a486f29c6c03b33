#!/bin/bash
# r06 s20: GRU recurrences with a continuous W ring (over unit blocks and time steps): parity tests, A/B against the
# previous kernels (tools/_trace/gru_old.so), per-kernel durations
O=gpurun_out/r6s20; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16.py tests/test_gpu_step_plan.py -x -q -m gpu -k "gru or GRU or launch_list" > $O/tests.txt 2>&1
tail -3 $O/tests.txt
for rep in 1 2; do
  RENET_HIP_LIB=tools/_trace/gru_old.so timeout 300 python bench.py --plain --steps 200 --warmup 20 > $O/bench_old_$rep.json 2> $O/bench_old_$rep.err
  timeout 300 python bench.py --plain --steps 200 --warmup 20 > $O/bench_new_$rep.json 2> $O/bench_new_$rep.err
done
for v in old new; do
  if [ $v = old ]; then export RENET_HIP_LIB=tools/_trace/gru_old.so; else unset RENET_HIP_LIB; fi
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_$v -o kt -- python $GRAFT_REPO_ROOT/bench.py --plain --steps 20 --warmup 3 > $GRAFT_REPO_ROOT/$O/prof_$v.log 2>&1)
  DB=$(find $O/prof_$v -name "*results.db" | head -1); python tools/prof_summary.py "$DB" $O/stats_$v.md 23
  grep -E "gru_|total kernel" $O/stats_$v.md
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6s20/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
