#!/bin/bash
# SQ-counter passes over single plane-format conv launches (tools/p3_pmc.py), one rocprofv3 --pmc run per counter set and shape.
#   gpurun -- 'bash tools/p3_pmc_set.sh <tag>'   -> gpurun_out/<tag>_pmc_sq.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1
out=$R/gpurun_out/${tag}_pmc_sq.txt; : > $out
while read -r what shape; do
  [ -z "$what" ] && continue
  echo "== $what $shape (N H Ci Co k s pad)" >> $out
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_MFMA" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
    rm -rf /tmp/pmc_$tag
    timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_$tag -- python $R/tools/p3_pmc.py $what $shape > /tmp/pmc_$tag.log 2>&1
    python $R/tools/agg_pmc.py /tmp/pmc_$tag p3_ >> $out 2>&1
  done
done <<SHAPES
fwd 16 64 64 256 1 1 0
fwd 16 32 128 128 3 1 1
fwd 16 16 256 1024 1 1 0
fwd 16 16 1024 256 1 1 0
bwdd 16 16 256 1024 1 1 0
bwdw 16 32 128 128 3 1 1
bwdw 16 16 1024 256 1 1 0
SHAPES
cat $out
