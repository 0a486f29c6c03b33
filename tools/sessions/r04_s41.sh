#!/bin/bash
# shape-dependent panel width (logits: 4): fabric reads per shape, launch times, parity
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=gpurun_out/r4s41
mkdir -p $O
export TMPDIR=/tmp
SHAPES="2048,23033,600,0,1 23033,600,2048,1,0 16000,600,800,0,1"
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/$O/p -o pmc -- python $R/tools/gemm_split_probe.py one base $SHAPES > $R/$O/p.log 2>&1)
DB=$(find $O/p -name "*results.db" | head -1)
(python tools/pmc_by_shape.py "$DB" FETCH_SIZE $SHAPES; timeout 300 python tools/gemm_split_probe.py one base $SHAPES 2>&1 | grep -v amdgpu.ids) > $O/by_shape.txt 2>&1
find $O -name "*.db" -delete
cat $O/by_shape.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config.py -x -q -m gpu -k "gemm or merged_pass" > $O/tests.txt 2>&1; tail -2 $O/tests.txt
