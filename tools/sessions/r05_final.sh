#!/bin/bash
# round 5: the profiles/ evidence at HEAD: whole GPU suite, smoke, the driver's bench command (compact line + bench_detail.json),
# rocprofv3 kernel stats + step timeline of the PLAIN pass only (bench.py --plain: the product configuration, no event-timed
# second pass), PMC traffic of the default mode, one-rank RCCL run of the reducer path
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=gpurun_out/r5final
mkdir -p $O
export TMPDIR=/tmp
(timeout 2400 python -m pytest tests -m gpu -q -s > $O/gpu_tests_full.txt 2>&1; grep -v amdgpu.ids $O/gpu_tests_full.txt | tail -4)
grep -v amdgpu.ids $O/gpu_tests_full.txt | grep -E "passed|failed|filtered MRR|full YAGO|pruned advance|relative L2|train-mode|paired|deviations|rank differences|predicted facts" > $O/gpu_tests.txt
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; wc -c $O/bench_driver_cmd.json
cp gpurun_out/bench_detail.json $O/bench_detail_driver_cmd.json
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err; wc -c $O/bench.json; cat $O/bench.json
cp gpurun_out/bench_detail.json $O/bench_detail.json
BENCH="python $R/bench.py --steps 20 --warmup 3 --plain"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt -- $BENCH > $R/$O/kt.log 2>&1)
DB=$(find $O/kt -name "*results.db" | head -1); python tools/prof_summary.py "$DB" $O/kernel_stats.md 20 && head -16 $O/kernel_stats.md
python tools/prof_timeline.py "$DB" $O/timeline.md
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d $R/$O/pmc_$C -o pmc -- python $R/bench.py --steps 3 --warmup 1 --plain > $R/$O/pmc_$C.log 2>&1)
done
F=$(find $O/pmc_FETCH_SIZE -name "*results.db" | head -1); W=$(find $O/pmc_WRITE_SIZE -name "*results.db" | head -1)
python tools/pmc_traffic.py "$F" "$W" $O/pmc_traffic_bf16x6.json bf16x6
find $O -name "*.db" -delete
RENET_FORCE_REDUCER=1 timeout 300 python bench.py --steps 20 --warmup 3 --plain > $O/bench_one_rank_rccl.json 2> $O/bench_one_rank_rccl.err; cat $O/bench_one_rank_rccl.json | cut -c1-400
