#!/bin/bash
# round 6, session 36: SQ / TCP counters of the GRU recurrences (old and new kernels) on tools/gru_bench.py
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD
O=gpurun_out/r6s36
mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --kernel-trace -d $R/$O/pmc1 -o pmc -- python $R/tools/gru_bench.py run $R/tools/_trace/gru_old.so $R/re-net_amd/csrc/librenet_hip.so > $R/$O/pmc1.log 2>&1)
DB=$(find $O/pmc1 -name "*results.db" | head -1); python tools/pmc_sq.py "$DB" $O/sq_gru.md gru_ > /dev/null; cat $O/sq_gru.md | cut -c1-260
# (a second pass with TCP_* counter names aborted in rocprofv3 -- names not valid on this build -- and is dropped)
rm -rf $O/pmc1
