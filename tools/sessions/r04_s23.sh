#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4s23
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_streams.py -m gpu -x -q > $O/t.log 2>&1; tail -8 $O/t.log
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16.py -m gpu -x -q -k "training_step or fast_mode or exact_fp32_mode or bf16_mode or global_model or gru" > $O/t2.log 2>&1; tail -4 $O/t2.log
B="--steps 100 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 0 --other-steps 0"
timeout 600 python bench.py $B > $O/bench.json 2> $O/bench.err; tail -c 200 $O/bench.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r4s23/bench.json').read().strip().splitlines()[-1])
print(round(j['value']), round(j['ms_per_step'],4), j.get('last_loss'))
PY
