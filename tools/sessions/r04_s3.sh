#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4s3
mkdir -p $O
timeout 600 python tools/gemm_f32_probe.py run > $O/probe.txt 2>&1
cat $O/probe.txt
