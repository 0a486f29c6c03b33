#!/bin/bash
# round 6, session 5: DMA pieces between MFMA groups, tile-shaped pack, XCD-aware softmax rows: tests, micro-benchmark (+ ablations), plain bench on / off
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6s5
mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_planes.py -x -q) > $O/planes_tests.log 2>&1; tail -5 $O/planes_tests.log
(timeout 300 python tools/planes_bench.py) > $O/planes_bench.log 2>&1; grep -v amdgpu.ids $O/planes_bench.log | cut -c1-200 | tail -9
for V in nodma nomfma; do
(RENET_HIP_LIB=$PWD/tools/_trace/p6_$V.so timeout 300 python tools/planes_bench.py --iters 10) > $O/bench_$V.log 2>&1; echo "== $V"; grep -v amdgpu.ids $O/bench_$V.log | cut -c1-170 | head -8
done
(timeout 300 python bench.py --plain --steps 20 --warmup 5) > $O/bench_planes.log 2>&1; grep -v amdgpu.ids $O/bench_planes.log | tail -1 | cut -c1-300
(RENET_PLANES=0 timeout 300 python bench.py --plain --steps 20 --warmup 5) > $O/bench_noplanes.log 2>&1; grep -v amdgpu.ids $O/bench_noplanes.log | tail -1 | cut -c1-300
