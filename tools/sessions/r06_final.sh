#!/bin/bash
# round 6, closing session: the whole GPU suite, the driver's bench command and the default one, rocprofv3 kernel trace +
# timeline of the plain pass, the two PMC passes (FETCH_SIZE / WRITE_SIZE) of the same command
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
O=gpurun_out/r6final
mkdir -p $O
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -x -q -rs) > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log
(timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3) > $O/bench_driver_cmd.log 2>&1; grep -v amdgpu.ids $O/bench_driver_cmd.log | tail -1 > $O/bench_line_driver_cmd.json; cut -c1-400 $O/bench_line_driver_cmd.json
(timeout 900 python bench.py) > $O/bench.log 2>&1; grep -v amdgpu.ids $O/bench.log | tail -1 > $O/bench_line.json; cut -c1-300 $O/bench_line.json
cp gpurun_out/bench_detail.json $O/bench_detail.json
BENCH="python $R/bench.py --steps 20 --warmup 3 --plain"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt -- $BENCH > $R/$O/kt.log 2>&1)
DB=$(find $O/kt -name "*results.db" | head -1); python tools/prof_summary.py "$DB" $O/kernel_stats.md 23 && head -12 $O/kernel_stats.md | cut -c1-160
python tools/prof_timeline.py "$DB" $O/timeline.md > /dev/null 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d $R/$O/pmc_$C -o pmc -- python $R/bench.py --steps 3 --warmup 1 --plain > $R/$O/pmc_$C.log 2>&1)
done
F=$(find $O/pmc_FETCH_SIZE -name "*results.db" | head -1); W=$(find $O/pmc_WRITE_SIZE -name "*results.db" | head -1)
python tools/pmc_traffic.py "$F" "$W" $O/pmc_traffic_bf16x6.json bf16x6
rm -rf $O/kt $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
