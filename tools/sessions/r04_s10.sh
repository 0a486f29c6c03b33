#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4s10
mkdir -p $O
timeout 600 python tools/infer_bench.py advance ICEWS18 > $O/advance.txt 2>&1; grep -v amdgpu.ids $O/advance.txt
RENET_GEMM=f16x3 timeout 600 python tools/infer_bench.py advance ICEWS18 > $O/advance_f16x3.txt 2>&1; grep -v amdgpu.ids $O/advance_f16x3.txt
timeout 600 python tools/advance_profile.py 14 > $O/advance_profile.txt 2>&1; grep -v amdgpu.ids $O/advance_profile.txt | head -36
timeout 900 python -m pytest tests/test_gpu_config.py tests/test_gpu_parity.py tests/test_gpu_e2e.py -m gpu -x -q -k "inference or eval or predict or yago_prefix" > $O/t.log 2>&1; tail -4 $O/t.log
timeout 600 python tools/infer_bench.py ICEWS18 3 200 > $O/stream.txt 2>&1; grep -v amdgpu.ids $O/stream.txt
