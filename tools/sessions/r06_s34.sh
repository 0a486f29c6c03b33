#!/bin/bash
# r06 s34: one feature split for the head (logits GEMM reads the ones-column planes): tests and A/B against the previous build
O=gpurun_out/r6s34; mkdir -p $O
R=$GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 1200 python -m pytest tests/test_gpu_step_plan.py tests/test_gpu_planes.py tests/test_gpu_streams.py tests/test_gpu_config.py -x -q -m gpu > $O/tests.txt 2>&1; tail -2 $O/tests.txt
for rep in 1 2 3; do
for v in old new; do
  if [ $v = old ]; then export RENET_HIP_LIB=$R/tools/_trace/prev.so; else unset RENET_HIP_LIB; fi
  timeout 300 python bench.py --plain --steps 200 --warmup 20 > $O/bench_${v}_$rep.json 2>/dev/null
done; done
unset RENET_HIP_LIB
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6s34/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d.get('plan_entries_per_step'))
    except Exception as e: print(f, 'ERR', e)
PY
