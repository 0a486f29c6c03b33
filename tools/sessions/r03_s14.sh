#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s14
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gather.py tests/test_gpu_bf16.py -m gpu -x -q > $O/t.log 2>&1; tail -4 $O/t.log
C5="--shape YAGO --hidden 400 --seq-len 15 --dtype bf16 --steps 60 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 0"
timeout 600 python bench.py $C5 > $O/bench_c5.json 2> $O/bench_c5.err
RENET_BF16_GATHER=0 timeout 600 python bench.py $C5 > $O/bench_c5_f32gather.json 2> $O/bench_c5_f32gather.err
python - <<'PY'
import json
for f in ('bench_c5','bench_c5_f32gather'):
    j=json.loads(open('gpurun_out/s14/%s.json' % f).read().strip().splitlines()[-1])
    print(f, j['value'], j['ms_per_step'], j.get('parity'))
    k=j['kernels']
    for n in sorted(k):
        if 'gather' in n: print('   ', n, k[n])
PY
