#!/bin/bash
# round 5, session 16: the unmodified train.py with the C list walker (3 epochs), the fused / builder / look-ahead GPU tests again
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5s16
mkdir -p $O
python -c "import sys; sys.path.insert(0, 're-net_amd'); import graph; print('listwalk:', graph._listwalk)"
timeout 300 python -m pytest tests/test_gpu_builder.py tests/test_gpu_parity.py -m gpu -x -q -k "list_api or fused or lookahead" > $O/tests.txt 2>&1; grep -v amdgpu.ids $O/tests.txt | tail -2
W=/tmp/refrun; rm -rf $W; mkdir -p $W/data/YAGO $W/models/YAGO
cp tools/_trace/refrun/data/YAGO/*.txt $W/data/YAGO/
python re-net_amd/preprocess.py $W/data/YAGO 10 > $O/preprocess.log 2>&1
cd $W
D=$R/tools/_trace/refrun
timeout 200 python $R/tools/run_reference_driver.py $D/pretrain.py -d YAGO --gpu 0 --dropout 0.5 --n-hidden 200 --lr 1e-3 --max-epochs 2 --batch-size 1024 > $O/pretrain.log 2>&1
timeout 400 python $R/tools/run_reference_driver.py $D/train.py -d YAGO --gpu 0 --dropout 0.5 --n-hidden 200 --lr 1e-3 --max-epochs 3 --batch-size 1024 --valid-every 5 > $O/train.log 2>&1
grep Epoch $O/train.log
