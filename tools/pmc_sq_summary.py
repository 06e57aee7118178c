"""One line per (shape, contraction kernel) from tools/p3_pmc_set.sh's output: python tools/pmc_sq_summary.py gpurun_out/<tag>_pmc_sq.txt"""
import sys, re, collections
cur = None; data = collections.OrderedDict()
for line in open(sys.argv[1]):
    if line.startswith('=='):
        cur = line.strip('= \n'); continue
    m = re.match(r'(.*?)\s+(SQ_\w+|TC\w+)\s+(\d+)\s+x(\d+)', line)
    if not m or cur is None: continue
    k = m.group(1).strip().replace('void ', '').replace('ldetr::', '')
    if not any(t in k for t in ('p3_nt', 'p3_c3', 'p3_tn')): continue
    data.setdefault((cur, k), {})[m.group(2)] = float(m.group(3))
print(f'{"shape / kernel":62s} {"CU-busy us":>10s} {"MFMA%":>6s} {"VALU/MFMA":>9s} {"SALU/MFMA":>9s} {"LDS/MFMA":>8s} {"VMEMrd/MFMA":>11s} {"bankconf%":>9s} {"waitLDS%":>8s} {"wait%":>6s} {"L2hit%":>6s} {"HBM rd MB":>9s} {"HBM wr MB":>9s}')
for (shape, k), d in data.items():
    g = lambda n: d.get(n, 0.0)
    mf = max(g('SQ_INSTS_MFMA'), 1.0); simd = max(g('SQ_BUSY_CU_CYCLES') * 4, 1.0)
    print(f'{(shape.split(" (")[0] + " " + k)[:62]:62s} {g("SQ_BUSY_CU_CYCLES") / 256 / 2400:10.1f} {100 * g("SQ_VALU_MFMA_BUSY_CYCLES") / simd:6.1f} {g("SQ_INSTS_VALU") / mf:9.1f} {g("SQ_INSTS_SALU") / mf:9.1f} '
          f'{g("SQ_INSTS_LDS") / mf:8.2f} {g("SQ_INSTS_VMEM_RD") / mf:11.2f} {100 * g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1):9.1f} {100 * g("SQ_WAIT_INST_LDS") / max(g("SQ_WAVE_CYCLES"), 1):8.1f} '
          f'{100 * g("SQ_WAIT_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1):6.1f} {100 * g("TCC_HIT_sum") / max(g("TCC_HIT_sum") + g("TCC_MISS_sum"), 1):6.1f} {g("TCC_EA0_RDREQ_sum") * 64 / 1e6:9.1f} {g("TCC_EA0_WRREQ_sum") * 64 / 1e6:9.1f}')
