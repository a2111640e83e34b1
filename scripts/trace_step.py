"""One step out of a rocprofv3 --kernel-trace of a training loop (the launches between two adam_kernel launches): launches, wall time,
kernel time per kernel name, busy / idle time per hardware queue. usage: trace_step.py <rocprof output dir> [rows] [marker]"""
import csv,glob,sys
from collections import defaultdict
def analyze(d, marker='adam', pick=-2):
    f=glob.glob(d+'/*/*_kernel_trace.csv')[0]
    rows=list(csv.DictReader(open(f)))
    names=[r['Kernel_Name'] for r in rows]
    ad=[i for i,n in enumerate(names) if marker in n]
    i0,i1=ad[pick-1],ad[pick]
    step=rows[i0+1:i1+1]
    t0=min(int(r['Start_Timestamp']) for r in step); t1=max(int(r['End_Timestamp']) for r in step)
    print(d,'launches',len(step),'wall us',(t1-t0)/1e3)
    agg=defaultdict(lambda:[0,0.0])
    for r in step:
        n=r['Kernel_Name'].split('(')[0][:80]
        agg[n][0]+=1; agg[n][1]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    for n,(c,t) in sorted(agg.items(),key=lambda kv:-kv[1][1])[:int(sys.argv[2]) if len(sys.argv)>2 else 25]:
        print(f"  {n:80s} {c:4d} {t:8.1f} us  avg {t/c:6.1f}")
    qs=defaultdict(list)
    for r in step: qs[r['Queue_Id']].append((int(r['Start_Timestamp']),int(r['End_Timestamp'])))
    for q,iv in qs.items():
        iv.sort(); busy=sum(e-s for s,e in iv)/1e3
        gaps=[(iv[i+1][0]-iv[i][1])/1e3 for i in range(len(iv)-1)]
        print('  queue',q,'launches',len(iv),'busy us',round(busy,1),'idle between own launches',round(sum(g for g in gaps if g>0),1))
analyze(sys.argv[1], sys.argv[3] if len(sys.argv)>3 else 'adam')
