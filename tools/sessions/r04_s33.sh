#!/bin/bash
# ablations of the bf16x6 two-phase k-loop (raw-buffer loader): no global loads / no LDS stores / no residual arithmetic / none of the three
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4s33
mkdir -p $O
(timeout 600 python tools/gemm_split_probe.py run) 2>&1 | grep -v amdgpu.ids > $O/ablate.txt
cat $O/ablate.txt
