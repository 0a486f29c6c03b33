#!/bin/bash
# round 4: refresh of the profiles/ evidence at HEAD: whole GPU suite, smoke, the default bench line (with companions and other
# configs), rocprofv3 kernel stats + step timeline + PMC traffic of the default mode
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=gpurun_out/r4final
mkdir -p $O
export TMPDIR=/tmp
(timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_tests.txt 2>&1; tail -4 $O/gpu_tests.txt)
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log)
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
BENCH="python $R/bench.py --steps 8 --warmup 2 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 0 --other-steps 0"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/kt -o kt -- $BENCH > $R/$O/kt.log 2>&1)
DB=$(find $O/kt -name "*results.db" | head -1); python tools/prof_summary.py "$DB" $O/kernel_stats.md 20 && head -14 $O/kernel_stats.md
python tools/prof_timeline.py "$DB" $O/timeline.md
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d $R/$O/pmc_$C -o pmc -- python $R/bench.py --steps 3 --warmup 1 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 0 --other-steps 0 > $R/$O/pmc_$C.log 2>&1)
done
F=$(find $O/pmc_FETCH_SIZE -name "*results.db" | head -1); W=$(find $O/pmc_WRITE_SIZE -name "*results.db" | head -1)
python tools/pmc_traffic.py "$F" "$W" $O/pmc_traffic_bf16x6.json bf16x6
find $O -name "*.db" -delete
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r4final/bench.json').read().strip().splitlines()[-1])
print(round(j['value']), round(j['ms_per_step'],4), j['gemm_mode'], 'frac', j['roofline']['frac'], 'traffic', j['roofline']['traffic'], j['roofline']['operand_bytes_per_launch'])
print('kernel_only', j['kernel_only']['ms_per_step'], j['kernel_only']['glue_ms_per_step'])
for k in ('value_f16x3','value_exact_f32'):
    r=j.get(k) or {}
    print(k, r.get('value'), r.get('ms_per_step'), (r.get('roofline') or {}).get('frac'), r.get('error'))
for k,r in (j.get('other_configs') or {}).items():
    p=r.get('parity') or {}
    print(k, r.get('value'), r.get('ms_per_step'), p.get('rel_err'), p.get('grad_rel_err'), (r.get('roofline') or {}).get('frac'), r.get('error'))
print('parity', j['parity']['rel_err'], j['parity']['grad_rel_err']); print('cpu', j['cpu_baseline']['value']); print('e2e dev', j['e2e_device_builder'], 'inline', j['e2e_inline'], 'workers', j['e2e_value'], 'enc', j['encoder_only']['value'])
for k,v in j['roofline_rgcn_gather'].items(): print(k, round(v['avg_us'],1), 'strict %.3f' % v['frac_strict'])
print(j['roofline_gru'])
PY
