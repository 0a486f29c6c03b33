#!/bin/bash
# Refreshes profiles/ evidence from HEAD: default bench line, rocprofv3 kernel-trace stats, PMC traffic passes,
# gather micro-bench, config 5 (bf16 storage) bench + kernel stats, the other dataset shapes.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=gpurun_out/prof3
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
BENCH="python $R/bench.py --steps 8 --warmup 2 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 0"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt -- $BENCH > $R/$O/kt.log 2>&1)
DB=$(find $O/kt -name "*results.db" | head -1); python tools/prof_summary.py "$DB" $O/kernel_stats.md 10 && head -12 $O/kernel_stats.md
python tools/prof_timeline.py "$DB" $O/timeline.md
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d $R/$O/pmc_$C -o pmc -- python $R/bench.py --steps 3 --warmup 1 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 0 > $R/$O/pmc_$C.log 2>&1)
done
F=$(find $O/pmc_FETCH_SIZE -name "*results.db" | head -1); W=$(find $O/pmc_WRITE_SIZE -name "*results.db" | head -1)
python tools/pmc_traffic.py "$F" "$W" $O/pmc_traffic.json
C5="--shape YAGO --hidden 400 --seq-len 15 --dtype bf16"
timeout 900 python bench.py $C5 --steps 100 --f32-steps 0 > $O/bench_c5.json 2> $O/bench_c5.err; tail -c 200 $O/bench_c5.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/kt5 -o kt -- $BENCH $C5 > $R/$O/kt5.log 2>&1)
DB=$(find $O/kt5 -name "*results.db" | head -1); python tools/prof_summary.py "$DB" $O/kernel_stats_c5.md 10 && head -12 $O/kernel_stats_c5.md
python tools/prof_timeline.py "$DB" $O/timeline_c5.md
for S in WIKI GDELT; do
  timeout 600 python bench.py --shape $S --steps 60 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 20 > $O/bench_$S.json 2> $O/bench_$S.err
done
(timeout 600 python tools/gather_bench.py both --json $O/gather_both.json) > $O/gather_both.log 2>&1; grep -v JSON $O/gather_both.log | tail -12
(timeout 600 python tools/gather_bench.py global --json $O/gather_global.json) > $O/gather_global.log 2>&1; grep -v JSON $O/gather_global.log | tail -8
python - <<'PY'
import json
for f in ('bench', 'bench_c5', 'bench_WIKI', 'bench_GDELT'):
    try:
        j = json.loads(open('gpurun_out/prof3/%s.json' % f).read().strip().splitlines()[-1])
        print(f, round(j['value']), round(j['ms_per_step'], 3), j['roofline']['frac'] if j.get('roofline') else None, j.get('parity'))
    except Exception as e:
        print(f, 'failed', e)
PY
find $O -name "*.db" -delete
(timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log)
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log)
