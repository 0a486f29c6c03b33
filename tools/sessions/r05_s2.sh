#!/bin/bash
# round 5, session 2: the reference's UNMODIFIED pretrain.py / train.py / test.py (shipped as untracked inputs under
# tools/_trace/refrun/, never committed) over the real HIP kernels, full YAGO, the README's commands (README.md:57-69) --
# plus --valid-every 5 (validations at epochs 10 / 15 / 20 instead of 11 of them: a CLI argument, not an edit).
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5s2
mkdir -p $O
W=/tmp/refrun; rm -rf $W; mkdir -p $W/data/YAGO $W/models/YAGO
cp tools/_trace/refrun/data/YAGO/*.txt $W/data/YAGO/
( time python re-net_amd/preprocess.py $W/data/YAGO 10 ) > $O/preprocess.log 2>&1; tail -3 $O/preprocess.log
cd $W
D=$R/tools/_trace/refrun
( time timeout 400 python $R/tools/run_reference_driver.py $D/pretrain.py -d YAGO --gpu 0 --dropout 0.5 --n-hidden 200 --lr 1e-3 --max-epochs 20 --batch-size 1024 ) > $O/pretrain.log 2>&1; grep -v amdgpu.ids $O/pretrain.log | tail -4
( time timeout 1200 python $R/tools/run_reference_driver.py $D/train.py -d YAGO --gpu 0 --dropout 0.5 --n-hidden 200 --lr 1e-3 --max-epochs 20 --batch-size 1024 --valid-every 5 ) > $O/train.log 2>&1; grep -v amdgpu.ids $O/train.log | tail -12
( time timeout 600 python $R/tools/run_reference_driver.py $D/test.py -d YAGO --gpu 0 --n-hidden 200 ) > $O/test.log 2>&1; grep -v amdgpu.ids $O/test.log | tail -12
ls -la $W/models/YAGO > $O/models_ls.txt
cd $R
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config.py tests/test_gpu_e2e.py -m gpu -x -q -s \
    -k "evaluate or inference or predict or yago_prefix or pruned" > $O/tests_pruned.txt 2>&1; grep -v amdgpu.ids $O/tests_pruned.txt | tail -6
