#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s22
mkdir -p $O
export RENET_GEMM_SKINNY=0
(RENET_H3_TALL=1 timeout 120 python tools/gemm_trace.py run 2048 23033 600 0 1 h3) > $O/trace_logits_tall.txt 2>&1; tail -12 $O/trace_logits_tall.txt
(RENET_H3_TALL=0 timeout 120 python tools/gemm_trace.py run 2048 23033 600 0 1 h3) > $O/trace_logits_128.txt 2>&1; tail -8 $O/trace_logits_128.txt
(RENET_H3_TALL=0 timeout 120 python tools/gemm_trace.py run 16000 600 800 0 1 h3) > $O/trace_gruin_128.txt 2>&1; tail -8 $O/trace_gruin_128.txt
(RENET_H3_TALL=0 timeout 120 python tools/gemm_trace.py run 23033 600 2048 1 0 h3) > $O/trace_dw_128.txt 2>&1; tail -8 $O/trace_dw_128.txt
(RENET_GEMM_TALL=0 timeout 120 python tools/gemm_trace.py run 16000 600 800 0 1 split) > $O/trace_gruin_x6.txt 2>&1; tail -6 $O/trace_gruin_x6.txt
