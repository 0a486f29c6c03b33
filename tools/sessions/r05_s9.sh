#!/bin/bash
# round 5, session 9: BASELINE configs[4] on its REAL dataset -- all of YAGO at n_hidden 400 / seq_len 15, bf16 storage vs the
# fp32-class bf16x6 mode of the same sizes, README schedule shortened to 10 epochs, three seeds, validation + test split; then the
# gather micro-benchmark at HEAD (kernels unchanged since round 4: a refresh of profiles/r04_gather_bench.txt)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5s9
mkdir -p $O
for M in bf16s bf16x6; do
RENET_GEMM=$M RENET_FULL_TEST=1 RENET_FULL_PRE_LR=1e-3 RENET_FULL_H=400 RENET_FULL_SEQ_LEN=15 timeout 900 python tools/yago_full_run.py 0.5 10 20 999 1000 1001 > $O/cfg5_$M.json 2> $O/cfg5_$M.err
grep -v amdgpu.ids $O/cfg5_$M.err | grep "seed [0-9]*:" | sed 's/"epoch_loss".*"seconds"/"seconds"/' | cut -c1-300
done
(timeout 300 python tools/gather_bench.py both --json $O/gather_both.json) > $O/gather_both.log 2>&1; grep -v "JSON\|amdgpu.ids" $O/gather_both.log | tail -12
(timeout 300 python tools/gather_bench.py global --json $O/gather_global.json) > $O/gather_global.log 2>&1; grep -v "JSON\|amdgpu.ids" $O/gather_global.log | tail -8
