#!/bin/bash
# round 5, session 15: where the 10.4 ms per step of the UNMODIFIED train.py loop go (cProfile of the driver over the HIP path, 2 epochs)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5s15
mkdir -p $O
W=/tmp/refrun; rm -rf $W; mkdir -p $W/data/YAGO $W/models/YAGO
cp tools/_trace/refrun/data/YAGO/*.txt $W/data/YAGO/
python re-net_amd/preprocess.py $W/data/YAGO 10 > $O/preprocess.log 2>&1
cd $W
D=$R/tools/_trace/refrun
timeout 200 python $R/tools/run_reference_driver.py $D/pretrain.py -d YAGO --gpu 0 --dropout 0.5 --n-hidden 200 --lr 1e-3 --max-epochs 2 --batch-size 1024 > $O/pretrain.log 2>&1
timeout 400 python -m cProfile -o $O/train.prof $R/tools/run_reference_driver.py $D/train.py -d YAGO --gpu 0 --dropout 0.5 --n-hidden 200 --lr 1e-3 --max-epochs 2 --batch-size 1024 --valid-every 5 > $O/train.log 2>&1
grep Epoch $O/train.log
python - <<PY
import pstats
st = pstats.Stats('$O/train.prof')
st.sort_stats('tottime').print_stats(32)
PY
