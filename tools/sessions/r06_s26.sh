#!/bin/bash
# r06 s26: new GRU kernels (forward: continuous stream; backward: one wave per unit block): tests, A/B bench, kernel stats
O=gpurun_out/r6s26; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16.py tests/test_gpu_step_plan.py tests/test_gpu_streams.py tests/test_gpu_config.py -x -q -m gpu > $O/tests.txt 2>&1
tail -3 $O/tests.txt
python tools/gru_bench.py run tools/_trace/gru_old.so re-net_amd/csrc/librenet_hip.so 2>&1 | grep -v amdgpu.ids | tee $O/gru_bench.txt
for rep in 1 2; do
for v in old new; do
  if [ $v = old ]; then export RENET_HIP_LIB=$R/tools/_trace/gru_old.so; else unset RENET_HIP_LIB; fi
  timeout 300 python bench.py --plain --steps 200 --warmup 20 > $O/bench_${v}_$rep.json 2> $O/bench_${v}_$rep.err
done; done
unset RENET_HIP_LIB
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_new -o kt -- python $R/bench.py --plain --steps 20 --warmup 3 > $R/$O/prof_new.log 2>&1)
DB=$(find $O/prof_new -name "*results.db" | head -1); python tools/prof_summary.py "$DB" $O/stats_new.md 23
grep -E "gru_|total kernel" $O/stats_new.md
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6s26/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
