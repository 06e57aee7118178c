for v in "X=1" "LDETR_P3_WSLOTS=256" "LDETR_P3_WSLOTS=768" "LDETR_P3_SLOTS=768"; do
  echo "== step $v"
  for rep in 1 2; do env $v python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
done
