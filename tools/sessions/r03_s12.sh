#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s12
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bf16.py -m gpu -x -q > $O/t_bf16.log 2>&1; tail -3 $O/t_bf16.log
C5="--shape YAGO --hidden 400 --seq-len 15 --dtype bf16 --steps 60 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 0"
timeout 600 python bench.py $C5 > $O/bench_c5.json 2> $O/bench_c5.err; tail -c 300 $O/bench_c5.err
timeout 600 python bench.py --dtype bf16 --steps 60 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 0 > $O/bench_bf16_d200.json 2> $O/bench_bf16_d200.err
python - <<'PY'
import json
for f in ('bench_c5','bench_bf16_d200'):
    j=json.loads(open('gpurun_out/s12/%s.json' % f).read().strip().splitlines()[-1])
    print(f, j['value'], j['ms_per_step'], j['gemm_mode'])
    for k,v in sorted(j['kernels'].items(), key=lambda kv:-kv[1]['ms_per_step'])[:4]: print('  %-28s %6.3f ms/step %s' % (k, v['ms_per_step'], {a:round(b,1) for a,b in v.items() if a in ('tflops','gbs','avg_us','calls_per_step')}))
PY
