python tools/p3_dev.py check checkb 2>&1 | grep -v amdgpu.ids | tail -16
python tools/p3_dev.py bench benchb 2>&1 | grep -v amdgpu.ids
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('value','ms_per_step') if k in d}); print({k:(v['ms_per_step'],v['tflops']) for k,v in d['roofline']['by_entry'].items() if k.startswith('p3')})"
