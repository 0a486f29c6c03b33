#!/bin/bash
# round 4, session 6: rocprofv3 kernel-trace stats + timeline + PMC traffic of the default (bf16x6) step; advance re-measured
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=gpurun_out/r4s6
mkdir -p $O
export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 8 --warmup 2 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 0 --other-steps 0"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt -- $BENCH > $R/$O/kt.log 2>&1)
DB=$(find $O/kt -name "*results.db" | head -1); python tools/prof_summary.py "$DB" $O/kernel_stats.md 20 && head -45 $O/kernel_stats.md
python tools/prof_timeline.py "$DB" $O/timeline.md
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d $R/$O/pmc_$C -o pmc -- python $R/bench.py --steps 3 --warmup 1 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 0 --other-steps 0 > $R/$O/pmc_$C.log 2>&1)
done
F=$(find $O/pmc_FETCH_SIZE -name "*results.db" | head -1); W=$(find $O/pmc_WRITE_SIZE -name "*results.db" | head -1)
python tools/pmc_traffic.py "$F" "$W" $O/pmc_traffic_bf16x6.json bf16x6
find $O -name "*.db" -delete
timeout 600 python tools/infer_bench.py advance ICEWS18 > $O/advance.txt 2>&1; cat $O/advance.txt
