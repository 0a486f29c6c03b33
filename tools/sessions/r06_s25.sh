#!/bin/bash
O=gpurun_out/r6s25; mkdir -p $O
python tools/gru_bench.py run tools/_trace/gru_old.so re-net_amd/csrc/librenet_hip.so tools/_trace/gru_bwd13.so 2>&1 | grep -v amdgpu.ids | tee $O/variants4.txt
