#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4s9
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "sparse_magnitude" > $O/t1.log 2>&1; grep -v amdgpu.ids $O/t1.log | tail -6
timeout 600 python tools/advance_profile.py 28 > $O/advance_profile.txt 2>&1; grep -v amdgpu.ids $O/advance_profile.txt | head -120
