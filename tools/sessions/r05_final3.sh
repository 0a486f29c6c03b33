#!/bin/bash
# round 5, closing session at HEAD: the whole GPU suite as the driver runs it, smoke, the default bench command
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5final3
mkdir -p $O
(timeout 2400 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; grep -v amdgpu.ids $O/gpu_tests.txt | tail -3)
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log)
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; wc -c $O/bench.json; cat $O/bench.json | cut -c1-900
cp gpurun_out/bench_detail.json $O/bench_detail.json
