#!/bin/bash
# usage: bash tools/p3_pmc.sh <tag> <what> [shape...]   -> gpurun_out/p3pmc_<tag>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1; shift
out=$R/gpurun_out/p3pmc_$tag.txt; : > $out
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_MISC"; do
  rm -rf /tmp/pmc_$tag
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_$tag -- python $R/tools/p3_pmc.py "$@" > /tmp/pmc_$tag.log 2>&1
  f=$(find /tmp/pmc_$tag -name '*counter_collection.csv' | head -1)
  python $R/tools/agg_pmc.py /tmp/pmc_$tag p3_ >> $out 2>&1
done
cat $out
