#!/bin/bash
# merged pass (both directions as one batch of 2B sequences): parity at config scale, then A/B against the paired step
mkdir -p gpurun_out/s5
timeout 1200 python -m pytest tests/test_gpu_config.py -m gpu -x -q -k "merged" > gpurun_out/s5/tests.log 2>&1
tail -5 gpurun_out/s5/tests.log
for mode in merged pair; do
  timeout 600 python bench.py --steps 100 --warmup 10 --cpu-steps 0 --f32-steps 0 --passes $mode > gpurun_out/s5/bench_$mode.json 2> gpurun_out/s5/bench_$mode.err
done
timeout 600 python bench.py --steps 60 --warmup 10 --cpu-steps 0 --f32-steps 0 --shape YAGO --hidden 400 --seq-len 15 --dtype bf16 --passes merged > gpurun_out/s5/bench_c5_merged.json 2> gpurun_out/s5/bench_c5_merged.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/s5/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d['value']), round(d['ms_per_step'],3), 'e2e', d.get('e2e_value'), d.get('e2e_inline'), 'gemm', d['kernels']['gemm_f32']['ms_per_step'], d['kernels']['gemm_f32']['tflops'])
    except Exception as e:
        print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-600:])
PY
