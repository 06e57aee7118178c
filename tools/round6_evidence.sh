#!/bin/bash
# Round-6 evidence in ONE gpurun call (everything lands in gpurun_out/<tag>/; what is judged is copied into profiles/ as r06<letter>_*):
#   the full -m gpu suite + smoke(); the default bench line WITH the per-(kernel, entry, shape) table (LDETR_ENGINE_SHAPES); rocprofv3 --kernel-trace --stats of
#   the same bench command; the two separate PMC passes (FETCH_SIZE / WRITE_SIZE) over tools/pmc_step.py; one replayed iteration in launch order at 16 and at
#   2 samples per GPU; BASELINE configs[4]'s per-GPU share (tools/evidence_cfg5.sh).
#   gpurun --timeout 3000 -- 'bash tools/round6_evidence.sh r06a'
tag=${1:-r06}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then
  timeout 1800 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -3 $out/pytest.log | cut -c1-300
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -2 $out/smoke.log | cut -c1-600
  ( tail -4 $out/pytest.log; tail -3 $out/smoke.log ) > $out/tests_smoke.txt
fi
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python tools/pmc_step.py > $out/pmc_$c.log 2>&1; echo "pmc $c rc=$?"
done
python tools/pmc_step.py --aggregate /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE 2 > $out/pmc_traffic.json 2> $out/pmc_agg.err
cp $out/pmc_traffic.json profiles/pmc_traffic.json       # the bench line below quotes it (same kernel sources: the digest is checked)
LDETR_ENGINE_SHAPES=$out/engine_shapes.txt timeout 900 python bench.py > $out/bench_line.json 2> $out/bench.err; echo "bench rc=$?"; cut -c1-300 $out/bench_line.json
rm -rf $out/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python bench.py --no-cpu-baseline --no-extra > $out/prof_bench.json 2> $out/prof_bench.err
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernel_stats.csv
rm -rf $out/prof
for b in 16 2; do
  rm -rf /tmp/prof_$b
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$b -- python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 6 --warmup 3 --per-gpu-batch $b > $out/prof_$b.log 2>&1
  f=$(find /tmp/prof_$b -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python tools/trace_order.py $f $out/launch_order_b$b.txt && grep '^# ' $out/launch_order_b$b.txt | head -2
done
bash tools/evidence_cfg5.sh $tag
