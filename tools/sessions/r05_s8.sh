#!/bin/bash
# round 5, session 8: the tests added after the final evidence session (per-model exact-fp32 mode, unmodified drivers as a GPU test,
# per-tensor bf16 gradient bounds, four-seed train-mode test), and the deterministic full-YAGO parity in the two remaining modes
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s8
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_streams.py tests/test_gpu_reference_drivers.py tests/test_gpu_bf16.py tests/test_gpu_e2e.py -m gpu -x -q -s \
   -k "exact_fp32_mode or unmodified or config5 or train_mode_filtered or two_models" > $O/tests.txt 2>&1
grep -v amdgpu.ids $O/tests.txt | grep -E "passed|failed|Error|error|train-mode|paired|unmodified drivers|assert" | cut -c1-400
for M in f16x3 bf16s; do
RENET_GEMM=$M timeout 300 python tools/yago_full_run.py 0.0 2 3 999 > $O/d0_$M.json 2> $O/d0_$M.err; grep -v amdgpu.ids $O/d0_$M.err | grep "seed 999:" | cut -c1-330
done
