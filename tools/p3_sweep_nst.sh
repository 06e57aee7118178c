#!/bin/bash
# Single-LDS-stage sweep of the plane-format gather kernel: gpurun -- 'bash tools/p3_sweep_nst.sh <tag>' -> gpurun_out/<tag>_nst_sweep.txt
out=gpurun_out/${1:-r04}_nst_sweep.txt; : > $out
for tile in 1 2 3 4; do for pf in 1 2; do for slots in 0 1024; do
  echo "== TILE=$tile PF=$pf SLOTS=$slots" >> $out
  if [ $slots = 0 ]; then unset LDETR_P3_SLOTS; else export LDETR_P3_SLOTS=$slots; fi
  LDETR_P3_NST=1 LDETR_P3_TILE=$tile LDETR_P3_PF=$pf P3_ONLY=1 python tools/p3_dev.py bench benchd 2>&1 | grep -v amdgpu.ids >> $out
done; done; done
