#!/bin/bash
# r06 s19: swizzled 64-byte-row LDS image of gemm_split_kernel: GEMM shapes and the training step, SWZ = 0 / 1 / 2
O=gpurun_out/r6s19; mkdir -p $O
for v in 0 1 2; do
  RENET_GEMM_SWZ=$v timeout 300 python tools/gemm_bench.py > $O/gemm_swz$v.txt 2>&1
done
RENET_GEMM_SWZ=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16.py -x -q -m gpu -k "gemm or split or skinny" > $O/tests_swz1.txt 2>&1
RENET_GEMM_SWZ=2 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16.py -x -q -m gpu -k "gemm or split or skinny" > $O/tests_swz2.txt 2>&1
for rep in 1 2; do
  for v in 0 1 2; do
    RENET_GEMM_SWZ=$v timeout 300 python bench.py --plain --steps 200 --warmup 20 > $O/bench_swz${v}_$rep.json 2> $O/bench_swz${v}_$rep.err
  done
done
tail -3 $O/tests_swz1.txt $O/tests_swz2.txt
for v in 0 1 2; do echo "== swz $v"; cat $O/gemm_swz$v.txt; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6s19/bench_swz*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
