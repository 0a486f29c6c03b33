#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/pp
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $R/gpurun_out/pp/a -o pmc -- python $R/tools/planes_one.py > $R/gpurun_out/pp/a.log 2>&1)
DB=$(find gpurun_out/pp/a -name "*results.db" | head -1)
python tools/pmc_summary.py "$DB" gemm_planes
python tools/pmc_summary.py "$DB" gemm_split_kernel
(cd /tmp && timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $R/gpurun_out/pp/b -o pmc -- python $R/tools/planes_one.py > $R/gpurun_out/pp/b.log 2>&1)
DB=$(find gpurun_out/pp/b -name "*results.db" | head -1)
python tools/pmc_summary.py "$DB" gemm_planes
python tools/pmc_summary.py "$DB" gemm_split_kernel
tail -3 gpurun_out/pp/a.log gpurun_out/pp/b.log
find gpurun_out/pp -name "*.db" -delete
