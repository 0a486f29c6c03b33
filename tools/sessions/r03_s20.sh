#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s20
mkdir -p $O
B="--steps 60 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 0"
for T in 200 400 0; do
RENET_H3_TALL=$T timeout 600 python bench.py $B > $O/bench_t$T.json 2> $O/bench_t$T.err
done
python - <<'PY'
import json
for f in ('bench_t200','bench_t400','bench_t0'):
    try:
        j=json.loads(open('gpurun_out/s20/%s.json' % f).read().strip().splitlines()[-1])
        print(f, round(j['value']), round(j['ms_per_step'],4), j['roofline']['achieved'])
        for g in j['gemm_shapes'][:12]: print('   ', g)
    except Exception as e:
        print(f, 'failed', e)
PY
timeout 2400 python -m pytest tests -m gpu -x -q > $O/t_all.log 2>&1; tail -5 $O/t_all.log
