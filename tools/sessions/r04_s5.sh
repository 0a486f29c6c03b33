#!/bin/bash
# round 4, session 5: inference advance with the relation broadcast (A/B), config-scale inference test, GPU tests of inference
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4s5
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_config.py tests/test_gpu_parity.py -m gpu -x -q -k "inference or eval or predict or topk or joint" -s > $O/t_infer.log 2>&1; tail -12 $O/t_infer.log
timeout 600 python tools/infer_bench.py advance ICEWS18 > $O/advance_bcast.txt 2>&1; cat $O/advance_bcast.txt
RENET_ADVANCE_BROADCAST=0 timeout 600 python tools/infer_bench.py advance ICEWS18 > $O/advance_rfold.txt 2>&1; cat $O/advance_rfold.txt
timeout 600 python tools/infer_bench.py ICEWS18 3 200 > $O/stream.txt 2>&1; cat $O/stream.txt
