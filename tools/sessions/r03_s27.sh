#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s27
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_streams.py -m gpu -x -q -k "segment_add" > $O/t.log 2>&1; tail -3 $O/t.log
B="--steps 100 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 0"
timeout 600 python bench.py --shape YAGO --hidden 400 --seq-len 15 --dtype bf16 $B > $O/bench_c5.json 2> $O/bench_c5.err
timeout 600 python bench.py $B > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
for f in ('bench_c5','bench'):
    j=json.loads(open('gpurun_out/s27/%s.json' % f).read().strip().splitlines()[-1])
    print(f, round(j['value']), round(j['ms_per_step'],4), j.get('last_loss'))
PY
