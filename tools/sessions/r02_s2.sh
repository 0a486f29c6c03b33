#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
(time timeout 2400 python -m pytest tests -x -q -m gpu) > gpurun_out/s2_tests.log 2>&1
echo "gpu tests rc=$?"; tail -6 gpurun_out/s2_tests.log
(time timeout 600 python tools/gather_bench.py sweep --json gpurun_out/s2_gather_sweep.json) > gpurun_out/s2_gather_sweep.log 2>&1
grep -E "items_fwd_full|items_bwdh_full|csr_" gpurun_out/s2_gather_sweep.log | grep -v JSON | head -40
(time timeout 900 python bench.py --f32-steps 0 --e2e-steps 0 --cpu-steps 1) > gpurun_out/s2_bench.json 2> gpurun_out/s2_bench.err
python - <<'PY'
import json
try:
    j = json.loads(open('gpurun_out/s2_bench.json').read().strip().splitlines()[-1])
    print({k: j[k] for k in ('value', 'ms_per_step', 'parity')})
    for k, v in j['roofline_rgcn_gather'].items(): print(k, round(v['avg_us'], 1), round(v['frac'], 3), round(v['frac_strict'], 3))
    for k, v in sorted(j['kernels'].items(), key=lambda kv: -kv[1]['ms_per_step']): print(k, v)
except Exception as e:
    print('bench parse failed', e); print(open('gpurun_out/s2_bench.err').read()[-3000:])
PY
