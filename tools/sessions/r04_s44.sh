#!/bin/bash
# the pruned inference advance at config scale (N_ent 23033, R 256, num_k 1000) against the reference's recorded facts
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4s44
RENET_ADVANCE_PRUNE=1 timeout 42 python -m pytest tests/test_gpu_config.py -x -q -s -m gpu -k "inference_at_config_scale and False" > gpurun_out/r4s44/t.log 2>&1
grep -v amdgpu.ids gpurun_out/r4s44/t.log | tail -8
