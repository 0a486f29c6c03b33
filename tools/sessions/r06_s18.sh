#!/bin/bash
# round 6, session 18: SQ counters of the planes kernel (LDS bank conflicts, MFMA busy, waits) on the micro-benchmark
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
O=gpurun_out/r6s18
mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $R/$O/pmc -o pmc -- python $R/tools/planes_bench.py --iters 3 > $R/$O/pmc.log 2>&1)
DB=$(find $O/pmc -name "*results.db" | head -1); python tools/pmc_sq.py "$DB" $O/sq_planes.md gemm > /dev/null; cat $O/sq_planes.md | cut -c1-260
rm -rf $O/pmc
