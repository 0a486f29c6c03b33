#!/bin/bash
# round 6, session 7: planes GEMM tile variants: 128 x 128 (two workgroups per CU, default) vs 256 x 128; tests; plain bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6s7
mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_planes.py -x -q) > $O/planes_tests.log 2>&1; tail -5 $O/planes_tests.log
for T in 128 256; do
(RENET_P6_TILE=$T timeout 300 python tools/planes_bench.py) > $O/planes_bench_$T.log 2>&1; echo "== tile $T"; grep -v amdgpu.ids $O/planes_bench_$T.log | cut -c1-150 | tail -9
done
for V in nodma nomfma; do
(RENET_HIP_LIB=$PWD/tools/_trace/p6_$V.so timeout 300 python tools/planes_bench.py --iters 10) > $O/bench_$V.log 2>&1; echo "== $V (tile 128)"; grep -v amdgpu.ids $O/bench_$V.log | cut -c1-150 | head -8
done
for T in 128 256; do
(RENET_P6_TILE=$T timeout 300 python bench.py --plain --steps 20 --warmup 5) > $O/bench_planes_$T.log 2>&1; grep -v amdgpu.ids $O/bench_planes_$T.log | tail -1 | cut -c100-220
done
(RENET_PLANES=0 timeout 300 python bench.py --plain --steps 20 --warmup 5) > $O/bench_noplanes.log 2>&1; grep -v amdgpu.ids $O/bench_noplanes.log | tail -1 | cut -c100-220
