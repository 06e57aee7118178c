#!/bin/bash
# Which kernels surround the ATen glue launches (copyBuffer, fills, adds, scalar multiplies) in the replayed step?  gpurun -- 'bash tools/trace_glue_context.sh [per_gpu_batch]'
export TMPDIR=/tmp; cd /tmp; R=$GRAFT_REPO_ROOT
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $R/bench.py --per-gpu-batch ${1:-16} --no-cpu-baseline --no-extra --no-roofline --steps 4 --warmup 2 > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1); python - "$f" <<'PY'
import csv,sys,collections,re
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
n=len(rows); seg=rows[int(n*0.75):]     # the last replayed iteration(s)
def short(n):
    n=n.replace('void ','').replace('ldetr::','').replace('at::native::','')
    n=re.sub(r'\(anonymous namespace\)::','',n)
    return n.split('(')[0][:64]
kinds={'copyBuffer':'copyBuffer','FillFunctor<float>':'fill','CUDAFunctor_add<float>':'add','AUnaryFunctor<float, float, float, binary_internal::MulFunctor':'mul_scalar','direct_copy':'copy_kernel','CatArray':'cat'}
for pat,tag in kinds.items():
    ctx=collections.Counter(); tot=0
    for i,r in enumerate(seg):
        if pat in r['Kernel_Name']:
            tot+=1
            prev=short(seg[i-1]['Kernel_Name']) if i>0 else ''; nxt=short(seg[i+1]['Kernel_Name']) if i+1<len(seg) else ''
            gs=r.get('Grid_Size', r.get('Grid_Size_X',''))
            ctx[(prev,nxt,gs)]+=1
    print(f'==== {tag}: {tot} launches in the last quarter of the trace')
    for k,v in ctx.most_common(18): print(f'{v:4d}  prev={k[0]:64s} next={k[1]:64s} grid={k[2]}')
PY
