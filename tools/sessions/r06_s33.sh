#!/bin/bash
O=gpurun_out/r6s33; mkdir -p $O
timeout 600 python tools/list_api_profile.py 45 2>&1 | grep -v amdgpu.ids > $O/list_api_profile.txt
head -75 $O/list_api_profile.txt | cut -c1-170
