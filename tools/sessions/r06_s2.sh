#!/bin/bash
# round 6, session 2: ablation builds of the planes GEMM (tools/p6_probe.py): DMA stream alone, MFMA stream alone
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6s2
mkdir -p $O
for V in nodma nomfma; do
(RENET_HIP_LIB=$PWD/tools/_trace/p6_$V.so timeout 300 python tools/planes_bench.py --iters 10) > $O/bench_$V.log 2>&1; echo "== $V"; grep -v amdgpu.ids $O/bench_$V.log | cut -c1-170 | head -8
done
