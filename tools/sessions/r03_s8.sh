#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s8
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "topk or joint_softmax" > $O/t_topk.log 2>&1; tail -12 $O/t_topk.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config.py -m gpu -x -q -s -k "evaluate or inference or shadowing or predict" > $O/t_inf.log 2>&1; tail -12 $O/t_inf.log
timeout 600 python tools/infer_bench.py > $O/infer_bench.log 2>&1; tail -8 $O/infer_bench.log
timeout 900 python tools/infer_bench.py advance > $O/advance.log 2>&1; tail -4 $O/advance.log
