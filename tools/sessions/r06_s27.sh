#!/bin/bash
# r06 s27: three-bucket reducer through a one-rank RCCL group; step-plan tests; bench with / without the forced reducer
O=gpurun_out/r6s27; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_step_plan.py tests/test_gpu_streams.py -x -q -m gpu > $O/tests.txt 2>&1
tail -15 $O/tests.txt
for b in 3 2; do
  (RENET_REDUCER_BUCKETS=$b RENET_FORCE_REDUCER=1 timeout 300 python bench.py --plain --steps 100 --warmup 10) > $O/rccl_b$b.log 2>$O/rccl_b$b.err
  tail -c 300 $O/rccl_b$b.err | grep -v amdgpu
done
timeout 300 python bench.py --plain --steps 100 --warmup 10 > $O/plain.log 2>/dev/null
python - <<'PY'
import json
for f in ('rccl_b3','rccl_b2','plain'):
    try:
        d=json.loads(open('gpurun_out/r6s27/%s.log'%f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
    except Exception as e: print(f,'ERR',e)
PY
