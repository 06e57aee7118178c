#!/bin/bash
# round-5 planning evidence: default bench, ordered kernel trace of one iteration at 16 and 2 samples per GPU, ATen glue sites
tag=${1:-r05a}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 600 python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 20 --warmup 5 > $out/bench16.json 2> $out/bench16.err; cut -c1-300 $out/bench16.json
timeout 600 python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 20 --warmup 5 --per-gpu-batch 2 > $out/bench2.json 2> $out/bench2.err; cut -c1-300 $out/bench2.json
for b in 16 2; do
  rm -rf /tmp/prof_$b
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$b -- python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 6 --warmup 3 --per-gpu-batch $b > $out/prof_$b.log 2>&1
  f=$(find /tmp/prof_$b -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python tools/trace_order.py $f $out/order_$b.txt && tail -60 $out/order_$b.txt | head -5
done
timeout 600 python tools/trace_aten.py 16 > $out/aten16.txt 2>&1; head -3 $out/aten16.txt
