#!/bin/bash
# bf16x6 operand split: residuals by v_dot2c_f32_bf16 vs unpack + packed subtraction -- bitwise comparison and timing
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4s32
mkdir -p $O
(timeout 300 python tools/gemm_split_probe.py cmp; timeout 300 python tools/gemm_split_probe.py run) 2>&1 | grep -v amdgpu.ids > $O/dot2.txt
cat $O/dot2.txt
