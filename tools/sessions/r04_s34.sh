#!/bin/bash
# two k-tiles of load lookahead in the bf16x6 two-phase kernels: bitwise comparison with one tile, timing, GEMM parity tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4s34
mkdir -p $O
(timeout 300 python tools/gemm_split_probe.py cmp; timeout 300 python tools/gemm_split_probe.py run) 2>&1 | grep -v amdgpu.ids > $O/la2.txt
cat $O/la2.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm" > $O/gemm_tests.txt 2>&1; tail -2 $O/gemm_tests.txt
