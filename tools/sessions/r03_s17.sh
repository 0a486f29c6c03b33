#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=gpurun_out/s17
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_streams.py tests/test_gpu_gather.py tests/test_gpu_parity.py tests/test_gpu_builder.py -m gpu -x -q > $O/t.log 2>&1; tail -6 $O/t.log
B="--steps 100 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 0"
timeout 600 python bench.py $B > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
C5="--shape YAGO --hidden 400 --seq-len 15 --dtype bf16"
timeout 600 python bench.py $C5 $B > $O/bench_c5.json 2> $O/bench_c5.err
BENCH="python $R/bench.py --steps 6 --warmup 2 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 0 --companions 0"
(cd /tmp && RENET_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt -- $BENCH > $R/$O/kt.log 2>&1)
DB=$(find $O/kt -name "*results.db" | head -1); python tools/prof_timeline.py "$DB" $O/timeline.md; python tools/prof_summary.py "$DB" $O/kernel_stats.md 8
find $O -name "*.db" -delete
python - <<'PY'
import json
for f in ('bench','bench_c5'):
    try:
        j=json.loads(open('gpurun_out/s17/%s.json' % f).read().strip().splitlines()[-1])
        print(f, round(j['value']), round(j['ms_per_step'],4), j.get('parity'), j.get('last_loss'))
    except Exception as e:
        print(f, 'failed', e)
PY
