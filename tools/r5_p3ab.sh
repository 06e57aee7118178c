#!/bin/bash
# A/B of the plane-format k-loop variants on ONE box: per-shape forward / paired-backward rates + the step
out=gpurun_out/${1:-r05i}; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_p3_gpu.py -x -q > $out/pytest_p3.log 2>&1; tail -2 $out/pytest_p3.log
for v in base early one prod; do
  if [ $v = prod ]; then unset LDETR_LIB; else export LDETR_LIB=$PWD/layoutdetr_amd/lib/variants/libldetr_hip_$v.so; fi
  P3_ONLY=1 timeout 600 python tools/p3_dev.py bench > $out/fwd_$v.txt 2>&1
  P3_ONLY=1 timeout 600 python tools/p3_dev.py pair > $out/pair_$v.txt 2>&1
  for rep in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])"; done
done
paste <(cut -c1-52 $out/fwd_base.txt) <(awk '{print $(NF-1), $NF}' $out/fwd_early.txt) <(awk '{print $(NF-1), $NF}' $out/fwd_one.txt) <(awk '{print $(NF-1), $NF}' $out/fwd_prod.txt) > $out/fwd_table.txt
cat $out/fwd_table.txt
