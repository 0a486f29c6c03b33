#!/bin/bash
# bf16x6 two-phase kernels on raw buffer descriptors (TileLoaderH): GEMM parity tests, then the step with RENET_GEMM_SPLIT_RAW=0 / 1
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4s30
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm" > $O/gemm_tests.txt 2>&1; tail -3 $O/gemm_tests.txt
B="python bench.py --steps 60 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 0 --other-steps 0"
for RAW in 0 1 0 1; do
RENET_GEMM_SPLIT_RAW=$RAW timeout 600 $B > $O/bench_raw$RAW.json 2> $O/bench_raw$RAW.err
python - <<PY
import json
j=json.loads(open('$O/bench_raw$RAW.json').read().strip().splitlines()[-1])
print('RAW=$RAW', round(j['value']), round(j['ms_per_step'],3), 'gemm avg %.1f us' % j['roofline']['avg_us'], 'frac %.3f' % j['roofline']['frac'], j['parity']['rel_err'], j['parity']['grad_rel_err'])
if '$RAW'=='1':
    for s in j['gemm_shapes'][:14]: print('   ', s['ta_tb_m_n_k_split'], s['avg_us'], s['tflops'])
PY
done
