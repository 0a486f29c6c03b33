#!/bin/bash
# round 4, session 7: C-store probe of the bf16x6 kernels, tall tile for dfeat, advance after the host-side clean-up, topk signed test,
# one-rank RCCL run with the reducer forced on
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4s7
mkdir -p $O
timeout 600 python tools/gemm_split_probe.py run > $O/split_probe.txt 2>&1; grep -v amdgpu.ids $O/split_probe.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config.py -m gpu -x -q -k "gemm or topk or inference" > $O/t.log 2>&1; tail -4 $O/t.log
timeout 600 python tools/infer_bench.py advance ICEWS18 > $O/advance.txt 2>&1; grep -v amdgpu.ids $O/advance.txt
B="--steps 60 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 0 --other-steps 0"
timeout 600 python bench.py $B > $O/bench.json 2> $O/bench.err
RENET_FORCE_REDUCER=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout 600 python bench.py $B > $O/bench_rccl1.json 2> $O/bench_rccl1.err; tail -c 300 $O/bench_rccl1.err
python - <<'PY'
import json
for f in ('bench','bench_rccl1'):
    try:
        j=json.loads(open('gpurun_out/r4s7/%s.json' % f).read().strip().splitlines()[-1])
        print(f, round(j['value']), round(j['ms_per_step'],4), j.get('last_loss'), j['roofline']['achieved'], j['roofline']['frac'])
        for g in j['gemm_shapes'][:6]: print('   ', g)
    except Exception as e:
        print(f, 'failed', e)
PY
