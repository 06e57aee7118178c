mkdir -p gpurun_out/r03c
for cfg in "--per-gpu-batch 2" "--per-gpu-batch 4" "--per-gpu-batch 4 --bg 512" "--text-mode encoder" "--text-mode encoder+lm"; do
  tag=$(echo "$cfg" | tr -d ' -' | tr '+' 'p')
  timeout 600 python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 20 --warmup 5 $cfg > gpurun_out/r03c/$tag.json 2> gpurun_out/r03c/$tag.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r03c/$tag.json').read().strip().splitlines()[-1]); print('$cfg', d['value'], d['ms_per_step'])
except Exception as e: print('$cfg FAILED', e)
PY
done
