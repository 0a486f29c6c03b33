#!/bin/bash
# round 5, session 6: full-YAGO deterministic parity (dropout 0, 2 epochs) of the product loop against the unmodified reference's fixture
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s6
mkdir -p $O
for M in bf16x6 f32; do
RENET_GEMM=$M timeout 600 python tools/yago_full_run.py 0.0 2 3 999 > $O/d0_$M.json 2> $O/d0_$M.err; grep -v amdgpu.ids $O/d0_$M.err | tail -4; cat $O/d0_$M.json
done
