#!/bin/bash
# round 3, session 1: new parity tests first, then the whole suite, bench line, kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=gpurun_out/s1
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_gather.py -m gpu -x -q -k "table_addressed" > $O/t_table.log 2>&1; tail -5 $O/t_table.log
timeout 1200 python -m pytest tests/test_gpu_config.py -m gpu -q -s -k "pretrain_scale or inference_at_config" > $O/t_scale.log 2>&1; tail -30 $O/t_scale.log
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -6 $O/gpu_tests.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err; python - <<'PY'
import json
j=json.loads(open('gpurun_out/s1/bench.json').read().strip().splitlines()[-1])
for k in ('value','ms_per_step','parity','encoder_only','kernel_only','roofline','cpu_baseline','e2e_inline','e2e_value'):
    print(k, j.get(k))
for k,v in sorted(j['kernels'].items(), key=lambda kv:-kv[1]['ms_per_step']): print('  %-28s %6.3f ms/step %s' % (k, v['ms_per_step'], {a:round(b,1) for a,b in v.items() if a in ('tflops','gbs','avg_us')}))
for g in j['gemm_shapes']: print('  ', g)
print(json.dumps(j['roofline_rgcn_gather'], indent=0))
PY
BENCH="python $R/bench.py --steps 8 --warmup 2 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 0"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt -- $BENCH > $R/$O/kt.log 2>&1)
DB=$(find $O/kt -name "*results.db" | head -1); echo "db: $DB"
python tools/prof_summary.py "$DB" $O/kernel_stats.md 10 && head -60 $O/kernel_stats.md
find $O -name "*.db" -delete
