#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=gpurun_out/s11
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/builder_bench.py 20 2>&1 | tail -3
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt -- python $R/tools/builder_bench.py 20 > $R/$O/kt.log 2>&1)
DB=$(find $O/kt -name "*results.db" | head -1); python tools/prof_summary.py "$DB" $O/builder_kernel_stats.md 23 && head -50 $O/builder_kernel_stats.md
find $O -name "*.db" -delete
