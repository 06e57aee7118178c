#!/bin/bash
# Tile x prefetch-depth sweep of the plane-format kernels: gpurun -- 'bash tools/p3_sweep_tiles.sh <tag>' -> gpurun_out/<tag>_tile_sweep.txt
out=gpurun_out/${1:-r04}_tile_sweep.txt; : > $out
for tile in 1 2 3 4; do for pf in 1 2 3; do
  echo "== TILE=$tile PF=$pf" >> $out
  LDETR_P3_TILE=$tile LDETR_P3_PF=$pf P3_ONLY=1 python tools/p3_dev.py bench benchd 2>&1 | grep -v amdgpu.ids >> $out
done; done
for wt in 1 2 3; do for pf in 1 2 3; do
  echo "== WTILE=$wt WPF=$pf" >> $out
  LDETR_P3_WTILE=$wt LDETR_P3_WPF=$pf P3_ONLY=1 python tools/p3_dev.py benchw 2>&1 | grep -v amdgpu.ids >> $out
done; done
