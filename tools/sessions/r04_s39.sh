#!/bin/bash
# fabric reads / writes per GEMM shape (bf16x6 kernels, XCD-aware tile order 8 and 0)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=gpurun_out/r4s39
mkdir -p $O
export TMPDIR=/tmp
SHAPES="2048,23033,600,0,1 2048,600,23033,0,0,6 23033,600,2048,1,0 16000,600,800,0,1 16000,400,600,0,0 600,800,16000,1,0,14 600,200,16000,1,0,39"
for ORD in 8 0; do
for C in FETCH_SIZE WRITE_SIZE; do
(cd /tmp && RENET_GEMM_TILE_ORDER=$ORD timeout 300 rocprofv3 --pmc $C --kernel-trace -d $R/$O/p_${ORD}_$C -o pmc -- python $R/tools/gemm_split_probe.py one base $SHAPES > $R/$O/p_${ORD}_$C.log 2>&1)
DB=$(find $O/p_${ORD}_$C -name "*results.db" | head -1)
echo "== tile order $ORD, $C"
python tools/pmc_by_shape.py "$DB" $C $SHAPES
done
done > $O/by_shape.txt 2>&1
find $O -name "*.db" -delete
cat $O/by_shape.txt
