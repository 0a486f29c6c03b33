#!/bin/bash
# GRU tile width A/B: tests with both widths, then bench at H=200 and config 5
mkdir -p gpurun_out/s4
timeout 900 python -m pytest tests -m gpu -x -q -k "gru or pair or training_step or zero_grad" > gpurun_out/s4/tests.log 2>&1
tail -5 gpurun_out/s4/tests.log
for mt in 16 32; do
  RENET_GRU_MT=$mt timeout 600 python bench.py --steps 100 --warmup 10 --cpu-steps 0 --f32-steps 0 > gpurun_out/s4/bench_h200_mt$mt.json 2> gpurun_out/s4/bench_h200_mt$mt.err
  RENET_GRU_MT=$mt timeout 600 python bench.py --steps 60 --warmup 10 --cpu-steps 0 --f32-steps 0 --shape YAGO --hidden 400 --seq-len 15 --dtype bf16 > gpurun_out/s4/bench_c5_mt$mt.json 2> gpurun_out/s4/bench_c5_mt$mt.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/s4/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d.get('roofline_gru'))
    except Exception as e:
        print(f, 'ERR', e)
PY
