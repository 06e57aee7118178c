mkdir -p gpurun_out/r03a; export TMPDIR=/tmp
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 4 --warmup 2 > /tmp/kt.log 2>&1
f=$(find /tmp/kt -name "*kernel_trace.csv" | head -1); python - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
print(len(rows), rows[0].keys())
# last 1/3 of the trace = steady-state replays
n=len(rows); seg=rows[int(n*0.6):]
ctx=collections.Counter()
def short(n): return n.replace('void ','').replace('ldetr::','').split('(')[0][:70]
for i,r in enumerate(seg):
    if 'copyBuffer' in r['Kernel_Name']:
        prev=short(seg[i-1]['Kernel_Name']) if i>0 else ''; nxt=short(seg[i+1]['Kernel_Name']) if i+1<len(seg) else ''
        dur=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000
        ctx[(prev,nxt, r.get('Stream_Id',''), r.get('Grid_Size',r.get('Grid_Size_X','')))]+=1
for k,v in ctx.most_common(40): print(v,k)
PY
