#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4s43
timeout 80 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm" > gpurun_out/r4s43/gemm_tests.txt 2>&1; tail -2 gpurun_out/r4s43/gemm_tests.txt
