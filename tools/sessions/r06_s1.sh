#!/bin/bash
# round 6, session 1: first run of the planes GEMM (csrc/gemm_p6.h): its tests, the head-shape micro-benchmark against the
# in-loop split, and the plain bench pass with the planes head on / off
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6s1
mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_planes.py -x -q) > $O/planes_tests.log 2>&1; tail -15 $O/planes_tests.log
(timeout 300 python tools/planes_bench.py) > $O/planes_bench.log 2>&1; grep -v amdgpu.ids $O/planes_bench.log | tail -12
(timeout 300 python bench.py --plain --steps 20 --warmup 5) > $O/bench_planes.log 2>&1; grep -v amdgpu.ids $O/bench_planes.log | tail -2 | cut -c1-600
(RENET_PLANES=0 timeout 300 python bench.py --plain --steps 20 --warmup 5) > $O/bench_noplanes.log 2>&1; grep -v amdgpu.ids $O/bench_noplanes.log | tail -2 | cut -c1-600
