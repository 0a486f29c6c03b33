#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s23
mkdir -p $O
export RENET_GEMM_SKINNY=0 RENET_FINE=1
(RENET_H3_TALL=0 timeout 120 python tools/gemm_trace.py run 2048 23033 600 0 1 h3) > $O/trace_logits_128.txt 2>&1; tail -7 $O/trace_logits_128.txt
(RENET_H3_TALL=1 timeout 120 python tools/gemm_trace.py run 2048 23033 600 0 1 h3) > $O/trace_logits_tall.txt 2>&1; tail -11 $O/trace_logits_tall.txt
