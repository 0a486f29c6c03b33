#!/bin/bash
# r06 s21: phase timeline of the persistent GRU forward kernel (continuous ring) with ablation builds, at full and low occupancy
O=gpurun_out/r6s21; mkdir -p $O
for V in "" _NOLOAD _NOMFMA; do
for B in 2048 128; do
  echo "== variant '$V' B=$B"
  RENET_TRACE_LIB=librenet_gru_trace$V.so timeout 200 python tools/gru_trace.py run 200 $B 10 2>&1 | grep -v amdgpu.ids | tee $O/trace$V_$B.txt
done; done
