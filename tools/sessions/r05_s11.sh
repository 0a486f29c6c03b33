#!/bin/bash
# round 5, session 11: the pipelined short-segment kernel of the segmented adds against the single-launch kernel
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s11
mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_streams.py -m gpu -x -q -k "segment_add" > $O/tests.txt 2>&1; grep -v amdgpu.ids $O/tests.txt | tail -3
RENET_SEGADD=one timeout 120 python tools/segadd_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/bench_one.txt
timeout 120 python tools/segadd_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/bench_pipelined.txt
