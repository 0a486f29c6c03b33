#!/bin/bash
# round 4, session 4: whole GPU suite at the new default (bf16x6) + the default bench line with companions
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4s4
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; tail -8 $O/gpu_tests.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r4s4/bench.json').read().strip().splitlines()[-1])
print(round(j['value']), round(j['ms_per_step'],4), j['gemm_mode'], j['roofline'])
print('kernel_only', j['kernel_only'])
print({n: (round(v['calls_per_step'],1), round(v['avg_us'],1), round(v['ms_per_step'],3)) for n,v in j['kernels'].items()})
for k in ('value_f16x3','value_exact_f32'):
    r=j.get(k) or {}
    print(k, r.get('value'), r.get('ms_per_step'), (r.get('roofline') or {}).get('frac'), r.get('error'))
for k,r in (j.get('other_configs') or {}).items():
    print(k, r.get('value'), r.get('ms_per_step'), (r.get('parity') or {}), (r.get('roofline') or {}).get('frac'), r.get('error'))
print('parity', j['parity']); print('cpu', j['cpu_baseline']['value']); print('e2e dev', j['e2e_device_builder'], 'enc', j['encoder_only'])
PY
