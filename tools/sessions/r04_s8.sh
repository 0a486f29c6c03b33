#!/bin/bash
# round 4, session 8: train-mode MRR per GEMM mode, softmax / f16x3-sparse / topk tests, advance, quick bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4s8
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "softmax or sparse_magnitude or topk or head or training_step" > $O/t1.log 2>&1; grep -v amdgpu.ids $O/t1.log | tail -12
timeout 2400 python -m pytest tests/test_gpu_e2e.py -m gpu -q -s -k "train_mode" > $O/t_train_mode.log 2>&1; grep -v amdgpu.ids $O/t_train_mode.log | tail -15
timeout 600 python tools/infer_bench.py advance ICEWS18 > $O/advance.txt 2>&1; grep -v amdgpu.ids $O/advance.txt
B="--steps 100 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 0 --other-steps 0"
timeout 600 python bench.py $B > $O/bench.json 2> $O/bench.err
RENET_SOFTMAX_REG=0 timeout 600 python bench.py $B > $O/bench_lds_softmax.json 2> $O/bench_lds_softmax.err
python - <<'PY'
import json
for f in ('bench','bench_lds_softmax'):
    try:
        j=json.loads(open('gpurun_out/r4s8/%s.json' % f).read().strip().splitlines()[-1])
        print(f, round(j['value']), round(j['ms_per_step'],4), j.get('last_loss'), j['roofline']['frac'], {n: round(v['avg_us'],1) for n,v in j['kernels'].items() if 'softmax' in n})
    except Exception as e:
        print(f, 'failed', e)
PY
