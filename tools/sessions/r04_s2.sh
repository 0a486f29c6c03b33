#!/bin/bash
# round 4, session 2: the buffer-addressed exact-fp32 GEMM kernel: parity tests, then the step in f32 mode
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4s2
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gemm" > $O/t_gemm.log 2>&1; tail -5 $O/t_gemm.log
B="--steps 40 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 0"
RENET_GEMM=f32 timeout 600 python bench.py $B > $O/bench_f32.json 2> $O/bench_f32.err; tail -c 300 $O/bench_f32.err
RENET_GEMM=f32 RENET_GEMM_TILE_ORDER=0 timeout 600 python bench.py $B > $O/bench_f32_noxcd.json 2> $O/bench_f32_noxcd.err
python - <<'PY'
import json
for f in ('f32','f32_noxcd'):
    try:
        j=json.loads(open('gpurun_out/r4s2/bench_%s.json' % f).read().strip().splitlines()[-1])
        print(f, round(j['value']), round(j['ms_per_step'],4), j.get('last_loss'), j['roofline']['achieved'])
        k=j['kernels']
        print('   ', {n: (round(v['calls_per_step'],1), round(v['avg_us'],1)) for n,v in k.items()})
        for g in j['gemm_shapes']: print('   ', g)
    except Exception as e:
        print(f, 'failed', e)
PY
