#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5s17
timeout 200 python tools/pretrain_probe.py 0.0 999 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5s17/pre.txt
timeout 300 python tools/pretrain_probe.py 0.5 999 1000 1001 1002 1003 1004 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r5s17/pre.txt
