#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=gpurun_out/s28
mkdir -p $O
export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 6 --warmup 2 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 0 --companions 0 --shape YAGO --hidden 400 --seq-len 15 --dtype bf16"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/kt5 -o kt -- $BENCH > $R/$O/kt5.log 2>&1)
DB=$(find $O/kt5 -name "*results.db" | head -1); python tools/prof_timeline.py "$DB" $O/timeline_c5.md; python tools/prof_summary.py "$DB" $O/kernel_stats_c5.md 8
find $O -name "*.db" -delete
grep segment_add $O/timeline_c5.md | awk -F'|' '{print $2,$4}'
