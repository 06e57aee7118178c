python tools/p3_dev.py check checkb 2>&1 | grep -v amdgpu.ids | tail -16
echo "=== patch kernel on"; python tools/p3_dev.py bench benchb 2>&1 | grep -v amdgpu.ids | grep "3x3"
echo "=== patch kernel off"; LDETR_P3_PATCH=0 python tools/p3_dev.py bench 2>&1 | grep -v amdgpu.ids | grep "3x3"
