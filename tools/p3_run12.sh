python tools/p3_dev.py check checkb 2>&1 | grep -v amdgpu.ids | tail -16
python tools/p3_dev.py bench benchb 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_model_gpu.py tests/test_composition_gpu.py -x -q -m gpu -k "training_iteration_vs_oracle or staged_backward or trunk_sharing or configs1 or forward_tuples or loss_phases" 2>&1 | tail -5
python bench.py --steps 10 --warmup 3 2>&1 | tail -1 > gpurun_out/p3_bench_1.json; python -c "
import json; d=json.loads(open('gpurun_out/p3_bench_1.json').read()); print({k: d[k] for k in ('value','ms_per_step') if k in d}); print(d.get('roofline',{}).get('by_entry'))"
