#!/bin/bash
# round 5, session 3: (a) tile-round quantization of the GRU-shaped GEMMs (625 tiles over 512 slots) against shapes that fill
# whole rounds; (b) fused directions through the real kernels + the unmodified train.py with RENET_FUSE_DIRECTIONS=1
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5s3
mkdir -p $O
timeout 200 python tools/gemm_split_probe.py one base 16000,600,800,0,1 16384,512,800,0,1 13056,640,800,0,1 16384,640,800,0,1 32768,512,800,0,1 \
   8192,512,800,0,1 16000,400,600,0,0 16384,384,600,0,0 16384,512,600,0,0 16000,600,600,0,0 23033,600,2048,1,0 32768,512,2048,1,0 2>&1 | grep -v amdgpu.ids > $O/quant.txt; cat $O/quant.txt
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused_directions" > $O/fused_test.txt 2>&1; tail -2 $O/fused_test.txt
W=/tmp/refrun; rm -rf $W; mkdir -p $W/data/YAGO $W/models/YAGO
cp tools/_trace/refrun/data/YAGO/*.txt $W/data/YAGO/
python re-net_amd/preprocess.py $W/data/YAGO 10 > $O/preprocess.log 2>&1
cd $W
D=$R/tools/_trace/refrun
timeout 200 python $R/tools/run_reference_driver.py $D/pretrain.py -d YAGO --gpu 0 --dropout 0.5 --n-hidden 200 --lr 1e-3 --max-epochs 2 --batch-size 1024 > $O/pretrain.log 2>&1
for F in 0 1; do
RENET_FUSE_DIRECTIONS=$F timeout 300 python $R/tools/run_reference_driver.py $D/train.py -d YAGO --gpu 0 --dropout 0.5 --n-hidden 200 --lr 1e-3 --max-epochs 3 --batch-size 1024 --valid-every 5 > $O/train_fuse$F.log 2>&1; grep -E "Epoch" $O/train_fuse$F.log
done
