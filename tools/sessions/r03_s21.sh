#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s21
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gemm" > $O/t_gemm.log 2>&1; tail -3 $O/t_gemm.log
B="--steps 100 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 0"
timeout 600 python bench.py $B > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
RENET_P32_WEIGHTS=0 timeout 600 python bench.py $B > $O/bench_nop32.json 2> $O/bench_nop32.err
python - <<'PY'
import json
for f in ('bench','bench_nop32'):
    try:
        j=json.loads(open('gpurun_out/s21/%s.json' % f).read().strip().splitlines()[-1])
        print(f, round(j['value']), round(j['ms_per_step'],4), j.get('last_loss'), j['roofline']['achieved'])
        k=j['kernels']
        print('   ', {n: (round(v['calls_per_step'],1), round(v['avg_us'],1)) for n,v in k.items()})
        for g in j['gemm_shapes'][:8]: print('   ', g)
    except Exception as e:
        print(f, 'failed', e)
PY
