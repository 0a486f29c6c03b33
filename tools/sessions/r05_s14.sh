#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s14
mkdir -p $O
timeout 300 python tools/host_profile.py 45 > $O/host_profile.txt 2>&1; grep -v amdgpu.ids $O/host_profile.txt | head -75
