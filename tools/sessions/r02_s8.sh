#!/bin/bash
# GRU step kernels: parity tests, then A/B against the persistent kernels
mkdir -p gpurun_out/s8
timeout 900 python -m pytest tests -m gpu -x -q -k "gru or pair or training_step or merged or zero_grad or bf16" > gpurun_out/s8/tests.log 2>&1
tail -8 gpurun_out/s8/tests.log
for mode in steps persistent; do
  RENET_GRU=$mode timeout 600 python bench.py --steps 100 --warmup 10 --cpu-steps 0 --f32-steps 0 --e2e-steps 0 > gpurun_out/s8/bench_h200_$mode.json 2> gpurun_out/s8/bench_h200_$mode.err
  RENET_GRU=$mode timeout 600 python bench.py --steps 60 --warmup 10 --cpu-steps 0 --f32-steps 0 --e2e-steps 0 --shape YAGO --hidden 400 --seq-len 15 --dtype bf16 > gpurun_out/s8/bench_c5_$mode.json 2> gpurun_out/s8/bench_c5_$mode.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/s8/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d['value']), round(d['ms_per_step'],3), d.get('roofline_gru'))
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-800:])
PY
