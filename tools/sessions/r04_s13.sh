#!/bin/bash
# what would a persistent fused bf16x6 kernel buy?  fused vs two-phase, with and without the C-store epilogue
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4s13
mkdir -p $O
for kern in split fused; do
  echo "== RENET_GEMM_KERNEL=$kern"
  RENET_GEMM_KERNEL=$kern RENET_GEMM_TALL=0 timeout 600 python tools/gemm_split_probe.py run 2>&1 | grep -v amdgpu.ids
done > $O/fused_vs_split.txt 2>&1
cat $O/fused_vs_split.txt
