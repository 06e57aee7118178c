#!/bin/bash
# default bench line (roofline + extras) + rocprofv3 kernel stats + ordered trace of one iteration
tag=${1:-r05p}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python bench.py > $out/bench_line.json 2> $out/bench.err; echo "bench rc=$?"; cut -c1-300 $out/bench_line.json
rm -rf $out/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 16 --warmup 4 > $out/prof_bench.json 2> $out/prof_bench.err
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernel_stats.csv
f=$(find $out/prof -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/trace_order.py $f $out/order_16.txt
rm -rf $out/prof
grep '^# ' $out/order_16.txt | head -2
