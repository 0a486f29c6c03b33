#!/bin/bash
# end-of-round evidence from HEAD: full GPU suite, smoke, default bench line, profiles
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/final/gpu_tests.log 2>&1; tail -4 gpurun_out/final/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.log 2>&1; tail -2 gpurun_out/final/smoke.log
timeout 900 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; tail -c 400 gpurun_out/final/bench.err
timeout 600 python bench.py --shape YAGO --hidden 400 --seq-len 15 --dtype bf16 --steps 100 --f32-steps 0 > gpurun_out/final/bench_c5.json 2> gpurun_out/final/bench_c5.err
bash tools/sessions/r02_profiles.sh > gpurun_out/final/profiles.log 2>&1; tail -30 gpurun_out/final/profiles.log
