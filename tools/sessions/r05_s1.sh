#!/bin/bash
# round 5, session 1: the driver's bench command with the compact line; the L2 rendezvous probe the weight-stationary GRU
# stands on; the pruned inference advance on a trained model and the inference tests with pruning on.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s1
mkdir -p $O
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; wc -c $O/bench.json; cat $O/bench.json
cp gpurun_out/bench_detail.json $O/bench_detail.json 2>/dev/null
mkdir -p tools/_trace && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probes/cluster_sync.hip -o tools/_trace/cluster_sync 2>/dev/null
timeout 30 tools/_trace/cluster_sync > $O/cluster_sync.txt 2>&1; cat $O/cluster_sync.txt
timeout 400 python tools/advance_pruned_bench.py ICEWS18 300 60 1000 200 > $O/advance_pruned.txt 2>&1; tail -12 $O/advance_pruned.txt
RENET_ADVANCE_PRUNE=1 timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config.py tests/test_gpu_e2e.py -m gpu -x -q -s \
    -k "evaluate or inference or predict or yago_prefix" > $O/tests_pruned.txt 2>&1; tail -6 $O/tests_pruned.txt
