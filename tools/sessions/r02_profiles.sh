#!/bin/bash
# Refreshes profiles/ evidence from HEAD: rocprofv3 kernel-trace stats of bench.py, PMC traffic passes, gather bench.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 8 --warmup 2 --cpu-steps 0 --e2e-steps 0 --f32-steps 0"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/kt -o kt -- $BENCH > $R/gpurun_out/prof/kt.log 2>&1)
DB=$(find gpurun_out/prof/kt -name "*results.db" | head -1); echo "db: $DB"
python tools/prof_summary.py "$DB" gpurun_out/prof/kernel_stats.md 10 && head -30 gpurun_out/prof/kernel_stats.md
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/prof/pmc_$C -o pmc -- python $R/bench.py --steps 3 --warmup 1 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 > $R/gpurun_out/prof/pmc_$C.log 2>&1)
done
F=$(find gpurun_out/prof/pmc_FETCH_SIZE -name "*results.db" | head -1); W=$(find gpurun_out/prof/pmc_WRITE_SIZE -name "*results.db" | head -1)
python tools/pmc_traffic.py "$F" "$W" gpurun_out/prof/pmc_traffic.json
(timeout 600 python tools/gather_bench.py batch --json gpurun_out/prof/gather_batch.json) > gpurun_out/prof/gather_batch.log 2>&1; cat gpurun_out/prof/gather_batch.log | grep -v JSON
(timeout 600 python tools/gather_bench.py both --json gpurun_out/prof/gather_both.json) > gpurun_out/prof/gather_both.log 2>&1; cat gpurun_out/prof/gather_both.log | grep -v JSON
(timeout 600 python tools/gather_bench.py global --json gpurun_out/prof/gather_global.json) > gpurun_out/prof/gather_global.log 2>&1; cat gpurun_out/prof/gather_global.log | grep -v JSON
# keep the merge small: drop the raw databases, keep summaries
find gpurun_out/prof -name "*.db" -size +20M -delete
