python -m pytest tests/test_model_gpu.py tests/test_composition_gpu.py -x -q -m gpu -k "training_iteration_vs_oracle or staged_backward or trunk_sharing or configs1 or loss_phases or two_rank or graph" 2>&1 | tail -5
for s in 1 0; do LDETR_P3_SIDE_WGRAD=$s python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('side', $s, {k: d[k] for k in ('value','ms_per_step') if k in d})"; done
