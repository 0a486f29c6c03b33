#!/bin/bash
# phase trace of the bf16x6 128-row two-phase kernel with the raw-buffer loader vs the generic one
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4s31
mkdir -p $O
for RAW in 1 0; do
for SH in "16000 600 800 0 1" "23033 600 2048 1 0" "2048 23033 600 0 1" "16000 600 600 0 0"; do
echo "== RAW=$RAW shape $SH"
(RENET_GEMM_SPLIT_RAW=$RAW RENET_GEMM_TALL=0 timeout 120 python tools/gemm_trace.py run $SH split) 2>&1 | grep -v amdgpu.ids | tail -9
done
done > $O/trace.txt 2>&1
cat $O/trace.txt
