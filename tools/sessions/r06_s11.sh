#!/bin/bash
# round 6, session 11: rocprofv3 kernel trace of the plain bench pass, C launch list on / off (per-kernel deltas)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
O=gpurun_out/r6s11
mkdir -p $O
export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 20 --warmup 3 --plain"
for P in 1 0; do
(cd /tmp && RENET_STEP_PLAN=$P timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/kt$P -o kt -- $BENCH > $R/$O/kt$P.log 2>&1)
DB=$(find $O/kt$P -name "*results.db" | head -1); python tools/prof_summary.py "$DB" $O/kernel_stats_plan$P.md 23
python tools/prof_timeline.py "$DB" $O/timeline_plan$P.md > /dev/null 2>&1
head -5 $O/kernel_stats_plan$P.md | cut -c1-200
rm -rf $O/kt$P
done
