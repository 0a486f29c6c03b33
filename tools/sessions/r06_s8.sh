#!/bin/bash
# round 6, session 8: the whole GPU suite and the default bench command with the planes head on
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6s8
mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -x -q) > $O/gpu_tests.log 2>&1; tail -6 $O/gpu_tests.log
(timeout 900 python bench.py) > $O/bench.log 2>&1; grep -v amdgpu.ids $O/bench.log | tail -1 | cut -c1-1500
cp gpurun_out/bench_detail.json $O/bench_detail.json 2>/dev/null
