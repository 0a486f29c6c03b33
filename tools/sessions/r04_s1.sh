#!/bin/bash
# round 4, session 1: per-shape GEMM tables of the three fp32-class modes at HEAD (which kernel the credited default should be)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4s1
mkdir -p $O
B="--steps 40 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 0"
for m in f32 bf16x6 f16x3; do
  RENET_GEMM=$m timeout 600 python bench.py $B > $O/bench_$m.json 2> $O/bench_$m.err; tail -c 300 $O/bench_$m.err
done
python - <<'PY'
import json
for f in ('f32','bf16x6','f16x3'):
    try:
        j=json.loads(open('gpurun_out/r4s1/bench_%s.json' % f).read().strip().splitlines()[-1])
        print(f, round(j['value']), round(j['ms_per_step'],4), j.get('last_loss'), j['roofline']['achieved'])
        k=j['kernels']
        print('   ', {n: (round(v['calls_per_step'],1), round(v['avg_us'],1)) for n,v in k.items()})
        for g in j['gemm_shapes']: print('   ', g)
    except Exception as e:
        print(f, 'failed', e)
PY
