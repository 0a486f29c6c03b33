#!/bin/bash
# NEXT ROUND, first session: the pruned inference advance -- timing and rows scored on a model trained on
# the synthetic ICEWS18-shaped stream, then every inference / evaluation test with the pruning on.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s1
mkdir -p $O
timeout 900 python tools/advance_pruned_bench.py ICEWS18 300 60 1000 200 > $O/advance_pruned.txt 2>&1; tail -12 $O/advance_pruned.txt
RENET_ADVANCE_PRUNE=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config.py tests/test_gpu_e2e.py -m gpu -x -q -s \
    -k "evaluate or inference or predict or yago_prefix" > $O/tests_pruned.txt 2>&1; tail -6 $O/tests_pruned.txt
# the number the weight-stationary GRU stands on: cycles per step of a G-workgroup rendezvous through L2
mkdir -p tools/_trace && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probes/cluster_sync.hip -o tools/_trace/cluster_sync 2>/dev/null
timeout 30 tools/_trace/cluster_sync > $O/cluster_sync.txt 2>&1; cat $O/cluster_sync.txt
