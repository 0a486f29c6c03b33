#!/bin/bash
# r06 s22: forward GRU with one wave per unit block (13 waves) + chunk ring through buffer loads
O=gpurun_out/r6s24; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16.py tests/test_gpu_step_plan.py tests/test_gpu_streams.py -x -q -m gpu -k "gru or GRU or launch_list or bit or identical" > $O/tests.txt 2>&1
tail -3 $O/tests.txt
timeout 200 python tools/gru_trace.py run 200 2048 10 2>&1 | grep -v amdgpu.ids | tee $O/trace.txt
R=$GRAFT_REPO_ROOT
for v in old new; do
  if [ $v = old ]; then export RENET_HIP_LIB=$R/tools/_trace/gru_old.so; else unset RENET_HIP_LIB; fi
  timeout 300 python bench.py --plain --steps 200 --warmup 20 > $O/bench_$v.json 2> $O/bench_$v.err
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$v -o kt -- python $R/bench.py --plain --steps 20 --warmup 3 > $R/$O/prof_$v.log 2>&1)
  DB=$(find $O/prof_$v -name "*results.db" | head -1); python tools/prof_summary.py "$DB" $O/stats_$v.md 23
  grep -E "gru_|total kernel" $O/stats_$v.md
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6s24/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
