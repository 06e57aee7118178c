#!/bin/bash
# full -m gpu suite + benches at 16 and 2 samples per GPU + ordered kernel trace of one iteration (round-5 working loop)
tag=${1:-r05}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -4 $out/pytest.log | cut -c1-300
fi
for b in 16 2; do
  timeout 600 python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 20 --warmup 5 --per-gpu-batch $b > $out/bench$b.json 2> $out/bench$b.err
  python - <<PY
import json
try:
    d=json.loads(open('$out/bench$b.json').read().strip().splitlines()[-1]); print('per-gpu batch $b:', d['value'], 'images/s', d['ms_per_step'], 'ms')
except Exception as e: print('bench $b FAILED', e)
PY
done
if [ -n "$TRACE" ]; then
for b in 16 2; do
  rm -rf /tmp/prof_$b
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$b -- python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 6 --warmup 3 --per-gpu-batch $b > $out/prof_$b.log 2>&1
  f=$(find /tmp/prof_$b -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python tools/trace_order.py $f $out/order_$b.txt && grep '^# ' $out/order_$b.txt | head -3
done
fi
