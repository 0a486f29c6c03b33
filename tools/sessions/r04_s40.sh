#!/bin/bash
# z-aware XCD tile order for split-K grids: fabric reads per shape, launch times, GEMM parity tests
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=gpurun_out/r4s40
mkdir -p $O
export TMPDIR=/tmp
SHAPES="2048,23033,600,0,1 2048,600,23033,0,0,6 23033,600,2048,1,0 16000,600,800,0,1 16000,400,600,0,0 600,800,16000,1,0,14 600,200,16000,1,0,39 600,600,16000,1,0,20 200,200,23033,1,0,82"
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/$O/p -o pmc -- python $R/tools/gemm_split_probe.py one base $SHAPES > $R/$O/p.log 2>&1)
DB=$(find $O/p -name "*results.db" | head -1)
(python tools/pmc_by_shape.py "$DB" FETCH_SIZE $SHAPES; timeout 300 python tools/gemm_split_probe.py one base $SHAPES 2>&1 | grep -v amdgpu.ids) > $O/by_shape.txt 2>&1
find $O -name "*.db" -delete
cat $O/by_shape.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm" > $O/gemm_tests.txt 2>&1; tail -2 $O/gemm_tests.txt
