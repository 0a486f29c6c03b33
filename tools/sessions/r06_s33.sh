#!/bin/bash
O=gpurun_out/r6s33; mkdir -p $O
timeout 600 python tools/list_api_profile.py 30 2>&1 | grep -v amdgpu.ids > $O/list_api_profile.txt
sed -n '/was called by/,$p' $O/list_api_profile.txt | cut -c1-200 | head -70
