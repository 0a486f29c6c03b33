#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4s14
mkdir -p $O
timeout 600 python tools/gemm_f32_probe.py run > $O/stagger.txt 2>&1
grep -v amdgpu.ids $O/stagger.txt
