export TMPDIR=/tmp
for pf in 1 2 3; do LDETR_P3_PF=$pf P3_ONLY=1 timeout 600 python tools/p3_dev.py bench 2>/dev/null | awk -v pf=$pf '{print "PF" pf, $0}' > gpurun_out/r05j_pf$pf.txt; done
paste <(cut -c1-60 gpurun_out/r05j_pf1.txt) <(awk '{print $(NF-1), $NF}' gpurun_out/r05j_pf1.txt) <(awk '{print $(NF-1), $NF}' gpurun_out/r05j_pf2.txt) <(awk '{print $(NF-1), $NF}' gpurun_out/r05j_pf3.txt)
