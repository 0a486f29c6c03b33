#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s13
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16.py -m gpu -x -q -k "gemm" > $O/t_gemm.log 2>&1; tail -3 $O/t_gemm.log
timeout 600 python bench.py --steps 100 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 0 > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
C5="--shape YAGO --hidden 400 --seq-len 15 --dtype bf16 --steps 60 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 0"
timeout 600 python bench.py $C5 > $O/bench_c5.json 2> $O/bench_c5.err
python - <<'PY'
import json
for f in ('bench','bench_c5'):
    j=json.loads(open('gpurun_out/s13/%s.json' % f).read().strip().splitlines()[-1])
    print(f, j['value'], j['ms_per_step'], j['kernels']['gemm_f32'])
    for g in j['gemm_shapes'][:8]: print('   ', g)
PY
