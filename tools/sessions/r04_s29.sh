#!/bin/bash
# config 5 (YAGO-shaped, n_hidden 400, seq_len 15, bf16 storage): bench line + kernel stats + timeline at HEAD
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=gpurun_out/r4s29
mkdir -p $O
export TMPDIR=/tmp
C5="--shape YAGO --hidden 400 --seq-len 15 --dtype bf16"
timeout 900 python bench.py $C5 --steps 100 --f32-steps 0 --other-steps 0 > $O/bench_c5.json 2> $O/bench_c5.err; tail -c 200 $O/bench_c5.err
BENCH="python $R/bench.py --steps 8 --warmup 2 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 0 --other-steps 0"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/kt5 -o kt -- $BENCH $C5 > $R/$O/kt5.log 2>&1)
DB=$(find $O/kt5 -name "*results.db" | head -1); python tools/prof_summary.py "$DB" $O/kernel_stats_c5.md 20 && head -16 $O/kernel_stats_c5.md
python tools/prof_timeline.py "$DB" $O/timeline_c5.md
find $O -name "*.db" -delete
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r4s29/bench_c5.json').read().strip().splitlines()[-1])
print(round(j['value']), round(j['ms_per_step'],3), j['gemm_mode'], j['roofline']['frac'], j['parity'])
print(j['roofline_gru'])
for k,v in j['roofline_rgcn_gather'].items(): print(k, round(v['avg_us'],1), 'strict %.3f' % v['frac_strict'])
PY
