#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=gpurun_out/r4s26
mkdir -p $O
export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 8 --warmup 2 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 0 --other-steps 0"
(cd /tmp && RENET_FORCE_REDUCER=1 timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt -- $BENCH > $R/$O/kt.log 2>&1)
DB=$(find $O/kt -name "*results.db" | head -1); python tools/prof_summary.py "$DB" $O/kernel_stats_rccl1.md 20 && grep -i "ccl\|allreduce\|AllReduce\|copy\|total kernel" $O/kernel_stats_rccl1.md | head -12
python tools/prof_timeline.py "$DB" $O/timeline_rccl1.md; grep -n -i "ccl\|reduce" $O/timeline_rccl1.md | head -12
find $O -name "*.db" -delete
