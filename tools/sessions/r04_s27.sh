#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 300 python tools/dp_overhead.py 60 2>&1 | grep -v "amdgpu.ids\|socket.cpp"
RENET_FORCE_REDUCER=1 timeout 300 python tools/dp_overhead.py 60 2>&1 | grep -v "amdgpu.ids\|socket.cpp"
