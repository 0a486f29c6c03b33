#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5s19
RENET_TEST_FULL_LENGTH=1 timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -x -q -s -k "full_length" > gpurun_out/r5s19/t.txt 2>&1; grep -v amdgpu.ids gpurun_out/r5s19/t.txt | grep -E "full-length|passed|failed|Error" | cut -c1-400
