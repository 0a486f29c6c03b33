#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4s24
mkdir -p $O
B="--steps 100 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 0 --other-steps 0"
timeout 600 python bench.py $B > $O/bench.json 2> $O/bench.err
RENET_FORCE_REDUCER=1 timeout 600 python bench.py $B > $O/bench_rccl1.json 2> $O/bench_rccl1.err; tail -c 400 $O/bench_rccl1.err
python - <<'PY'
import json
for f in ('bench','bench_rccl1'):
    try:
        j=json.loads(open('gpurun_out/r4s24/%s.json' % f).read().strip().splitlines()[-1])
        print(f, round(j['value']), round(j['ms_per_step'],4), j.get('last_loss'), 'ranks seen', j.get('rccl_ranks_seen'))
    except Exception as e:
        print(f, 'failed', e)
PY
