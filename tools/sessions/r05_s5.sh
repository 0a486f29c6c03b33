#!/bin/bash
# round 5, session 5: the reference's UNMODIFIED pretrain.py / train.py / test.py, README commands verbatim (README.md:57-69,
# eleven validations included), over the HIP path with the list-API accelerations of this round (merged directions on the
# device builder, look-ahead evaluation: tools/run_reference_driver.py defaults); then the same test.py with the look-ahead off.
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5s5
mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lookahead or evaluate_filter" > $O/tests.txt 2>&1; grep -v amdgpu.ids $O/tests.txt | tail -3
W=/tmp/refrun; rm -rf $W; mkdir -p $W/data/YAGO $W/models/YAGO
cp tools/_trace/refrun/data/YAGO/*.txt $W/data/YAGO/
( time python re-net_amd/preprocess.py $W/data/YAGO 10 ) > $O/preprocess.log 2>&1; tail -3 $O/preprocess.log
cd $W
D=$R/tools/_trace/refrun
( time timeout 400 python $R/tools/run_reference_driver.py $D/pretrain.py -d YAGO --gpu 0 --dropout 0.5 --n-hidden 200 --lr 1e-3 --max-epochs 20 --batch-size 1024 ) > $O/pretrain.log 2>&1; grep -v amdgpu.ids $O/pretrain.log | tail -4
( time timeout 1200 python $R/tools/run_reference_driver.py $D/train.py -d YAGO --gpu 0 --dropout 0.5 --n-hidden 200 --lr 1e-3 --max-epochs 20 --batch-size 1024 ) > $O/train.log 2>&1; grep -v amdgpu.ids $O/train.log | grep -E "Epoch|MRR|real" 
( time timeout 600 python $R/tools/run_reference_driver.py $D/test.py -d YAGO --gpu 0 --n-hidden 200 ) > $O/test.log 2>&1; grep -v amdgpu.ids $O/test.log | tail -10
( time RENET_LOOKAHEAD_EVAL=0 timeout 600 python $R/tools/run_reference_driver.py $D/test.py -d YAGO --gpu 0 --n-hidden 200 ) > $O/test_nolook.log 2>&1; grep -v amdgpu.ids $O/test_nolook.log | tail -10
