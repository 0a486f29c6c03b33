#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s16
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_streams.py tests/test_gpu_config.py -m gpu -x -q > $O/t.log 2>&1; tail -6 $O/t.log
B="--steps 100 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 0"
timeout 600 python bench.py $B > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
RENET_SIDE_STREAM=0 timeout 600 python bench.py $B > $O/bench_noside.json 2> $O/bench_noside.err
RENET_SIDE_STREAM=0 RENET_DUAL_HEAD=0 timeout 600 python bench.py $B > $O/bench_old.json 2> $O/bench_old.err
C5="--shape YAGO --hidden 400 --seq-len 15 --dtype bf16 --steps 60 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 0"
timeout 600 python bench.py $C5 > $O/bench_c5.json 2> $O/bench_c5.err
python - <<'PY'
import json
for f in ('bench','bench_noside','bench_old','bench_c5'):
    try:
        j=json.loads(open('gpurun_out/s16/%s.json' % f).read().strip().splitlines()[-1])
        print(f, round(j['value']), round(j['ms_per_step'],4), j.get('parity'), j.get('last_loss'))
    except Exception as e:
        print(f, 'failed', e)
PY
