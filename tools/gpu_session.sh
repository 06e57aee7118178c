#!/bin/bash
# Development aid: one gpurun call = the targeted tests of the current work + short bench runs; logs land in gpurun_out/.
#   gpurun --timeout 1500 -- 'bash tools/gpu_session.sh <tag> [pytest -k expression] [variants...]'
tag=${1:-s}; kexpr=${2:-}; shift 2 || true
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
if [ -n "$kexpr" ]; then
  timeout 1200 python -m pytest tests -m gpu -x -q -k "$kexpr" -s > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
  tail -25 $out/pytest.log
fi
for v in "$@"; do
  if [ "$v" = prod ]; then unset LDETR_LIB; else export LDETR_LIB=$PWD/layoutdetr_amd/lib/variants/libldetr_hip_$v.so; fi
  for rep in 1 2; do
    timeout 300 python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 20 --warmup 5 > $out/bench_${v}_$rep.json 2> $out/bench_${v}_$rep.err
    python - <<PY
import json
try:
    d=json.loads(open('$out/bench_${v}_$rep.json').read().strip().splitlines()[-1]); print('$v', $rep, d['value'], d['ms_per_step'])
except Exception as e:
    print('$v', $rep, 'FAILED', e)
PY
  done
done
