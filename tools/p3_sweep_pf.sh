#!/bin/bash
# Prefetch-depth sweep of the plane-format gather kernel: gpurun -- 'bash tools/p3_sweep_pf.sh <tag>' -> gpurun_out/<tag>_pf_sweep.txt
out=gpurun_out/${1:-r04}_pf_sweep.txt; : > $out
for pf in 1 2 3 4; do
  echo "== LDETR_P3_PF=$pf" >> $out
  LDETR_P3_PF=$pf python tools/p3_dev.py bench benchb 2>&1 | grep -v amdgpu.ids >> $out
done
for pf in 1 2 3 4; do
  echo "== LDETR_P3_PF=$pf step" >> $out
  LDETR_P3_PF=$pf python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" >> $out
done
