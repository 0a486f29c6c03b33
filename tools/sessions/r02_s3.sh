#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_bf16.py tests/test_gpu_config.py -x -q -m gpu -s -k "bf16 or pair" 2>&1 | tail -6
echo "== torchrun N=1 with forced reducer"
RENET_FORCE_REDUCER=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 50 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 2>gpurun_out/s3_torchrun.err | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('torchrun', round(j['value']), round(j['ms_per_step'],3), j['last_loss'])"
tail -3 gpurun_out/s3_torchrun.err
echo "== plain"
python bench.py --steps 50 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain', round(j['value']), round(j['ms_per_step'],3), j['last_loss'])"
