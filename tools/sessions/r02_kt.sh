#!/bin/bash
# kernel-trace stats only (quick look at the step's composition)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/kt
export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 8 --warmup 2 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 $BENCH_EXTRA"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt/kt -o kt -- $BENCH > $R/gpurun_out/kt/kt.log 2>&1)
DB=$(find gpurun_out/kt/kt -name "*results.db" | head -1); echo "db: $DB"
python tools/prof_summary.py "$DB" gpurun_out/kt/kernel_stats.md 10 && head -60 gpurun_out/kt/kernel_stats.md
find gpurun_out/kt -name "*.db" -delete
