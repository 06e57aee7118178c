python tools/p3_dev.py check checkb 2>&1 | grep -v amdgpu.ids | tail -16
for t in 1 2 3; do echo "=== TILE $t"; LDETR_P3_TILE=$t python tools/p3_dev.py bench 2>&1 | grep -v amdgpu.ids; done
