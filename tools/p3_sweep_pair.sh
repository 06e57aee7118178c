#!/bin/bash
# Paired-backward sweep: gpurun -- 'bash tools/p3_sweep_pair.sh <tag>' -> gpurun_out/<tag>_pair_sweep.txt
out=gpurun_out/${1:-r04}_pair_sweep.txt; : > $out
for order in 0; do for slots in 512 1024; do   # (the block-order switch was removed after the sweep: weight gradient first)
  echo "== ORDER=$order WSLOTS=$slots" >> $out
  LDETR_P3_PAIR_ORDER=$order LDETR_P3_WSLOTS=$slots python tools/p3_dev.py pair 2>&1 | grep -v amdgpu.ids | cut -c1-25,85-200 >> $out
done; done
for wnst in 1 2; do for wpf in 1 2; do
  echo "== WNST=$wnst WPF=$wpf" >> $out
  LDETR_P3_WNST=$wnst LDETR_P3_WPF=$wpf P3_ONLY=1 python tools/p3_dev.py benchw 2>&1 | grep -v amdgpu.ids >> $out
done; done
for v in "LDETR_P3_PAIR=0" "LDETR_P3_PAIR=1 LDETR_P3_PAIR_ORDER=0" "LDETR_P3_PAIR=1 LDETR_P3_PAIR_ORDER=1" "LDETR_P3_PAIR=3 LDETR_P3_PAIR_ORDER=0" "LDETR_P3_PAIR=0 LDETR_P3_WNST=1"; do
  echo "== step $v" >> $out
  env $v python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> $out
done
