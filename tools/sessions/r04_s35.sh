#!/bin/bash
# kernel timeline of a step of bench.py's FIRST pass (product configuration: no per-kernel events, side streams on)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=gpurun_out/r4s35
mkdir -p $O
export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 8 --warmup 2 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 0 --other-steps 0"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $R/$O/kt -o kt -- $BENCH > $R/$O/kt.log 2>&1)
DB=$(find $O/kt -name "*results.db" | head -1)
python - <<PY
import sqlite3
c=sqlite3.connect("$DB").cursor()
print([r[1] for r in c.execute("pragma table_info(kernels)").fetchall()])
PY
python tools/prof_timeline.py "$DB" $O/timeline_product.md adam_kernel 5
python tools/prof_timeline.py "$DB" $O/timeline_product6.md adam_kernel 6
find $O -name "*.db" -delete
