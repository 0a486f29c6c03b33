#!/bin/bash
# GPU session 1 (round 2): new parity + gather tests, gather sweep, bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests/test_gpu_gather.py -x -q -m gpu) > gpurun_out/s1_gather_tests.log 2>&1
echo "gather tests rc=$?" 
tail -5 gpurun_out/s1_gather_tests.log
(time timeout 600 python tools/gather_bench.py sweep --json gpurun_out/s1_gather_sweep.json) > gpurun_out/s1_gather_sweep.log 2>&1
echo "sweep rc=$?"
grep -E "items_fwd_full|items_bwdh_full|csr_" gpurun_out/s1_gather_sweep.log | head -60
(time timeout 900 python -m pytest tests/test_gpu_config.py -x -q -m gpu) > gpurun_out/s1_config_tests.log 2>&1
echo "config tests rc=$?"
tail -15 gpurun_out/s1_config_tests.log
(time timeout 900 python bench.py) > gpurun_out/s1_bench.json 2> gpurun_out/s1_bench.err
echo "bench rc=$?"
python - <<'PY'
import json
try:
    j = json.loads(open('gpurun_out/s1_bench.json').read().strip().splitlines()[-1])
    print({k: j[k] for k in ('value', 'ms_per_step', 'parity', 'value_exact_f32', 'roofline', 'roofline_gru', 'cpu_baseline', 'e2e_value', 'e2e_inline')})
    print(json.dumps(j['roofline_rgcn_gather'], indent=1))
    for k, v in sorted(j['kernels'].items(), key=lambda kv: -kv[1]['ms_per_step']): print(k, v)
except Exception as e:
    print('bench parse failed', e); print(open('gpurun_out/s1_bench.err').read()[-3000:])
PY
(time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_e2e.py -x -q -m gpu) > gpurun_out/s1_parity_tests.log 2>&1
echo "parity tests rc=$?"
tail -15 gpurun_out/s1_parity_tests.log
