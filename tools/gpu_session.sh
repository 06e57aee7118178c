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
if [ -n "$PROF" ]; then
  cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
  rm -rf gpurun_out/$tag/prof
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/prof -- python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 16 --warmup 4 $BENCH_ARGS > $out/prof_bench.log 2>&1
  f=$(find gpurun_out/$tag/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernel_stats.csv
  find gpurun_out/$tag/prof -type f ! -name "*kernel_stats.csv" -delete
  tail -2 $out/prof_bench.log | cut -c1-300
fi
for v in "$@"; do
  vv=$(echo "$v" | tr ":=," "___")
  unset LDETR_LIB; for e in $CLEAR_ENVS; do unset $e; done; CLEAR_ENVS=""
  case "$v" in
    prod) ;;
    env:*) for kv in $(echo "${v#env:}" | tr "," " "); do export "$kv"; CLEAR_ENVS="$CLEAR_ENVS ${kv%%=*}"; done ;;
    *) export LDETR_LIB=$PWD/layoutdetr_amd/lib/variants/libldetr_hip_$v.so ;;
  esac
  for rep in 1 2; do
    timeout 300 python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 20 --warmup 5 $BENCH_ARGS > $out/bench_${vv}_$rep.json 2> $out/bench_${vv}_$rep.err
    python - <<PY
import json
try:
    d=json.loads(open('$out/bench_${vv}_$rep.json').read().strip().splitlines()[-1]); print('$v', $rep, d['value'], d['ms_per_step'])
except Exception as e:
    print('$v', $rep, 'FAILED', e)
PY
  done
done
