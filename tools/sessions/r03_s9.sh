#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s9
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_builder.py -m gpu -x -q > $O/t_builder.log 2>&1; tail -30 $O/t_builder.log
