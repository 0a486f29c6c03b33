#!/bin/bash
# round 5, session 13: score-head parameters first in the flat buffers (two collectives per step): the tests that pin HipAdam's arithmetic,
# then the one-rank RCCL step again
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s13
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_streams.py tests/test_gpu_e2e.py tests/test_gpu_builder.py -m gpu -x -q -k "not train_mode and not full_yago" > $O/tests.txt 2>&1; grep -v amdgpu.ids $O/tests.txt | tail -3
for i in 1 2; do RENET_FORCE_REDUCER=1 timeout 200 python bench.py --steps 40 --warmup 5 --plain 2>/dev/null | grep -o '"value":[0-9.]*,"unit":"triples/s","n_gpus":1,"steps":40,"warmup":5,"ms_per_step":[0-9.]*'; done
timeout 200 python bench.py --steps 40 --warmup 5 --plain 2>/dev/null | grep -o '"value":[0-9.]*,"unit":"triples/s","n_gpus":1,"steps":40,"warmup":5,"ms_per_step":[0-9.]*'
