#!/bin/bash
# round 5, last evidence session at HEAD: the whole GPU suite (all fixtures present), smoke, the driver's bench command, the unmodified
# drivers with the README commands once more (C list walker in)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5final2
mkdir -p $O
(timeout 2400 python -m pytest tests -m gpu -q -s > $O/gpu_tests_full.txt 2>&1; grep -v amdgpu.ids $O/gpu_tests_full.txt | tail -3)
grep -v amdgpu.ids $O/gpu_tests_full.txt | grep -E "passed|failed|filtered MRR|full YAGO|criterion|pruned advance|relative L2|train-mode|paired|deviations|unmodified drivers|predicted facts" > $O/gpu_tests.txt
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; wc -c $O/bench_driver_cmd.json; cut -c1-700 $O/bench_driver_cmd.json
cp gpurun_out/bench_detail.json $O/bench_detail_driver_cmd.json
W=/tmp/refrun; rm -rf $W; mkdir -p $W/data/YAGO $W/models/YAGO
cp tools/_trace/refrun/data/YAGO/*.txt $W/data/YAGO/
python re-net_amd/preprocess.py $W/data/YAGO 10 > $O/preprocess.log 2>&1
cd $W
D=$R/tools/_trace/refrun
( time timeout 400 python $R/tools/run_reference_driver.py $D/pretrain.py -d YAGO --gpu 0 --dropout 0.5 --n-hidden 200 --lr 1e-3 --max-epochs 20 --batch-size 1024 ) > $O/pretrain.log 2>&1; grep real $O/pretrain.log
( time timeout 1200 python $R/tools/run_reference_driver.py $D/train.py -d YAGO --gpu 0 --dropout 0.5 --n-hidden 200 --lr 1e-3 --max-epochs 20 --batch-size 1024 ) > $O/train.log 2>&1; grep -v amdgpu.ids $O/train.log | grep -E "Epoch 00(01|10|20)|valid MRR|real" | tail -16
( time timeout 600 python $R/tools/run_reference_driver.py $D/test.py -d YAGO --gpu 0 --n-hidden 200 ) > $O/test.log 2>&1; grep -v amdgpu.ids $O/test.log | grep -E "Hits|MRR|real"
