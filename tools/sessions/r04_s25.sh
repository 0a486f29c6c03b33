#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4s25
mkdir -p $O
B="--steps 30 --cpu-steps 0 --e2e-steps 0 --f32-steps 0 --enc-steps 0 --other-steps 0"
PYTHONFAULTHANDLER=1 RENET_FORCE_REDUCER=1 timeout 600 python bench.py $B > $O/bench_rccl1.json 2> $O/bench_rccl1.err; echo "rc=$?"
grep -v amdgpu.ids $O/bench_rccl1.err | tail -40
head -c 300 $O/bench_rccl1.json
