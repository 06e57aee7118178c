#!/bin/bash
# BASELINE configs[4] evidence on ONE GPU's share (4 samples per GPU, 512x512, encoder+lm, texts padded to 256): bench line, kernel stats, PMC traffic.
#   gpurun --timeout 1500 -- 'bash tools/evidence_cfg5.sh <tag>'      -> gpurun_out/<tag>/cfg5_*
tag=${1:-cfg5}; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
ARGS="--bg 512 --per-gpu-batch 4 --text-mode encoder+lm"
timeout 600 python bench.py $ARGS --steps 10 --warmup 3 --no-cpu-baseline > $out/cfg5_bench_line.json 2> $out/cfg5_bench.err
tail -c 400 $out/cfg5_bench.err
cd /tmp; cd - > /dev/null
rm -rf $out/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python bench.py $ARGS --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra > $out/cfg5_prof_bench.log 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/cfg5_kernel_stats.csv
rm -rf $out/prof
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $out/pmc_$c
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -- python tools/pmc_step.py --b 4 --bg 512 --text-mode encoder+lm > $out/cfg5_pmc_$c.log 2>&1
done
python tools/pmc_step.py --aggregate $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE 2 > $out/cfg5_pmc_traffic.json 2> $out/cfg5_pmc_agg.err
rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
python - <<PY
import json
d=json.loads(open("$out/cfg5_bench_line.json").read().strip().splitlines()[-1])
print("cfg5", d["value"], d["ms_per_step"], d["config"]["workload"][:200])
r=d["roofline"]; print({k:r[k] for k in ("kernel","achieved","peak","frac","launches_per_step","avg_us")})
t=json.load(open("$out/cfg5_pmc_traffic.json")); print("pmc engine GB/iter", (t["engine_total"]["fetch"]+t["engine_total"]["write"])/1e9, "launches", t["engine_total"]["launches"])
PY
