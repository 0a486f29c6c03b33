#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s29
mkdir -p $O
(timeout 300 python tools/infer_bench.py ICEWS18 3 200) > $O/infer.txt 2>&1; tail -4 $O/infer.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "f16x3_gemm_bounds" > $O/t.log 2>&1; tail -2 $O/t.log
