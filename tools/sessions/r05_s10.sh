#!/bin/bash
# round 5, session 10: train-mode full-YAGO parity, the seeds the reference fixture holds so far
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s10
mkdir -p $O
SEEDS=$(python -c "import numpy as np; print(' '.join(str(int(s)) for s in np.load('tests/golden/e2e_yago_full_drop.npz')['seeds']))")
for M in bf16x6; do
RENET_GEMM=$M timeout 600 python tools/yago_full_run.py 0.5 3 3 $SEEDS > $O/drop_$M.json 2> $O/drop_$M.err; grep -v amdgpu.ids $O/drop_$M.err | grep "seed [0-9]*:" | cut -c1-700
done
