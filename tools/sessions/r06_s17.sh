#!/bin/bash
# round 6, session 17: ablation: the planes k-loop without DMA and without fragment reads (MFMAs + barriers only)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6s17
mkdir -p $O
for V in nodma noread; do
(RENET_HIP_LIB=$PWD/tools/_trace/p6_$V.so timeout 300 python tools/planes_bench.py --iters 10) > $O/bench_$V.log 2>&1; echo "== $V"; grep -v amdgpu.ids $O/bench_$V.log | cut -c1-150 | head -7
done
