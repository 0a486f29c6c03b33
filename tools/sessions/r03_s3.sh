#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s3
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "skinny" > $O/t_skinny.log 2>&1; tail -5 $O/t_skinny.log
timeout 600 python bench.py --steps 100 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 20 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
RENET_GEMM_SKINNY=0 timeout 600 python bench.py --steps 100 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 20 > $O/bench_noskinny.json 2> $O/bench_ns.err
python - <<'PY'
import json
for f in ('bench', 'bench_noskinny'):
    j=json.loads(open('gpurun_out/s3/%s.json' % f).read().strip().splitlines()[-1])
    print(f, j['value'], j['ms_per_step'], 'enc', j['encoder_only']['ms_per_step'], 'gemm', j['kernels']['gemm_f32'])
    for g in j['gemm_shapes']:
        if g['ta_tb_m_n_k_split'][3] <= 256 and g['ta_tb_m_n_k_split'][4] <= 256: print('   ', g)
PY
