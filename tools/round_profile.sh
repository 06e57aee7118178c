#!/bin/bash
# Round-end evidence in ONE gpurun call: the full -m gpu suite, smoke(), the default bench line, rocprofv3 --kernel-trace --stats of the
# same bench command, and the two separate PMC passes (FETCH_SIZE / WRITE_SIZE) over tools/pmc_step.py.  Everything lands in
# gpurun_out/<tag>/; copy what is to be judged into profiles/.
#   gpurun --timeout 2400 -- 'bash tools/round_profile.sh r03a'
tag=${1:-r}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
if [ -z "$SKIP_TESTS" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -3 $out/pytest.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
fi
timeout 900 python bench.py > $out/bench_line.json 2> $out/bench.err; echo "bench rc=$?"; cut -c1-400 $out/bench_line.json
rm -rf $out/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python bench.py --no-cpu-baseline --no-extra > $out/prof_bench.json 2> $out/prof_bench.err
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernel_stats.csv
rm -rf $out/prof
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python tools/pmc_step.py > $out/pmc_$c.log 2>&1; echo "pmc $c rc=$?"
done
python tools/pmc_step.py --aggregate /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE 2 > $out/pmc_traffic.json 2> $out/pmc_agg.err; head -c 600 $out/pmc_traffic.json
