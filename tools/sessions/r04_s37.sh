#!/bin/bash
# 256-row tile forced on / off for the step's shapes, bf16x6 two-phase kernels with the raw-buffer loader
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4s37
mkdir -p $O
for T in 0 1; do
echo "== RENET_GEMM_TALL=$T (0: never, 1: always)"
RENET_GEMM_TALL=$T timeout 300 python tools/gemm_split_probe.py one base 2>&1 | grep -v amdgpu.ids
done > $O/tall.txt 2>&1
cat $O/tall.txt
