#!/bin/bash
# round 6, session 6: rocprofv3 kernel trace of the plain bench pass with the planes head (and without, for the per-kernel delta)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
O=gpurun_out/r6s6
mkdir -p $O
export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 20 --warmup 3 --plain"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt -- $BENCH > $R/$O/kt.log 2>&1)
DB=$(find $O/kt -name "*results.db" | head -1); python tools/prof_summary.py "$DB" $O/kernel_stats_planes.md 23 && head -40 $O/kernel_stats_planes.md | cut -c1-150
(cd /tmp && RENET_PLANES=0 timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/kt0 -o kt -- $BENCH > $R/$O/kt0.log 2>&1)
DB=$(find $O/kt0 -name "*results.db" | head -1); python tools/prof_summary.py "$DB" $O/kernel_stats_noplanes.md 23 && head -24 $O/kernel_stats_noplanes.md | cut -c1-150
rm -rf $O/kt $O/kt0
