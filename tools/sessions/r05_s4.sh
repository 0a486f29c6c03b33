#!/bin/bash
# round 5, session 4: the list API on the device builder (bit-identity tests), then the unmodified train.py with and without it
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5s4
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_builder.py tests/test_gpu_parity.py -m gpu -x -q -k "list_api or fused" > $O/tests.txt 2>&1; grep -v amdgpu.ids $O/tests.txt | tail -12
W=/tmp/refrun; rm -rf $W; mkdir -p $W/data/YAGO $W/models/YAGO
cp tools/_trace/refrun/data/YAGO/*.txt $W/data/YAGO/
python re-net_amd/preprocess.py $W/data/YAGO 10 > $O/preprocess.log 2>&1
cd $W
D=$R/tools/_trace/refrun
timeout 200 python $R/tools/run_reference_driver.py $D/pretrain.py -d YAGO --gpu 0 --dropout 0.5 --n-hidden 200 --lr 1e-3 --max-epochs 2 --batch-size 1024 > $O/pretrain.log 2>&1
for V in "0 1" "1 0" "1 1"; do
set -- $V
RENET_FUSE_DIRECTIONS=$1 RENET_DEVICE_BUILDER_LISTS=$2 timeout 300 python $R/tools/run_reference_driver.py $D/train.py -d YAGO --gpu 0 --dropout 0.5 --n-hidden 200 --lr 1e-3 --max-epochs 3 --batch-size 1024 --valid-every 5 > $O/train_fuse$1_dev$2.log 2>&1; echo "fuse=$1 device_builder=$2"; grep -E "Epoch|Error|error" $O/train_fuse$1_dev$2.log | tail -4
done
