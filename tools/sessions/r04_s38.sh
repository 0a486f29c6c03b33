#!/bin/bash
# SQ counters of the bf16x6 GEMM kernels on the step's shapes (MFMA pipe occupancy, wait / issue split, LDS conflicts)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=gpurun_out/r4s38
mkdir -p $O
export TMPDIR=/tmp
CMD="python $R/tools/gemm_split_probe.py one base 2048,23033,600,0,1 23033,600,2048,1,0 16000,600,800,0,1 16000,600,600,0,0 4096,4096,4096,0,1"
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d $R/$O/pmc -o pmc -- $CMD > $R/$O/pmc.log 2>&1)
tail -3 $O/pmc.log
DB=$(find $O/pmc -name "*results.db" | head -1)
python tools/pmc_sq.py "$DB" $O/sq_gemm.md gemm_split
find $O -name "*.db" -delete
