#!/bin/bash
# where the GPU time of the simulated 8-rank z-slab run goes: per-kernel sums of ONE step (rocprofv3 kernel trace of bench.py --workload slab --sim-ranks 8)
#   slab_kernel_sums.sh [ranks] [native]      native: the C++ driver (csrc/sharded.hip) instead of 3dsift_amd/slab.py
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
NATIVE=""; export S3D_NSTEPS=3
if [ "$2" = "native" ]; then NATIVE="--native --no-rank-times"; export S3D_NSTEPS=2; fi
rm -rf /tmp/p_ss; rocprofv3 --kernel-trace -d /tmp/p_ss --output-format csv -- python3 $R/bench.py --workload slab $NATIVE --sim-ranks ${1:-8} --steps 1 --warmup 1 > /tmp/p_ss.out 2>/dev/null
f=$(find /tmp/p_ss -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys,collections,os
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# bench.py runs the workload three times (first call, warm-up, timed step): a step begins with its ranks' level-0 launches
# (k_march_level<2, ...), take everything from the first of the last step's on
s3=[r for r in rows if 's3d::' in r['Kernel_Name']]
l0=[i for i,r in enumerate(s3) if 'k_march_level<2' in r['Kernel_Name']]
nsteps=int(os.environ.get("S3D_NSTEPS","3"))
last=s3[l0[len(l0)-len(l0)//nsteps]:]
# (the native driver also launches the base blur in its constructor-free warm-up the same number of times per step: the split by count holds)
t0=int(last[0]['Start_Timestamp']); t1=max(int(r['End_Timestamp']) for r in last)
tot=collections.defaultdict(float); cnt=collections.Counter()
busy=0; cur_end=0
ev=sorted((int(r['Start_Timestamp']),int(r['End_Timestamp'])) for r in last)
for s,e in ev:
    if s>cur_end: busy+=e-s; cur_end=e
    elif e>cur_end: busy+=e-cur_end; cur_end=e
for r in last:
    n=r['Kernel_Name'].split('(')[0].replace('void ','').replace('s3d::','')
    tot[n]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3; cnt[n]+=1
print("one step: wall %.2f ms, GPU busy (union of kernel intervals) %.2f ms, sum of kernel durations %.2f ms, %d launches"%((t1-t0)/1e6,busy/1e6,sum(tot.values())/1e3,len(last)))
for n,v in sorted(tot.items(),key=lambda x:-x[1])[:24]: print("  %-34s %5d launches %9.1f us"%(n,cnt[n],v))
# per queue (the sharded stages run on the ranks' shared stream, the replicated tails on the tail contexts' streams)
perq=collections.defaultdict(lambda: collections.defaultdict(float))
for r in last:
    n=r['Kernel_Name'].split('(')[0].replace('void ','').replace('s3d::','')
    perq[r.get('Queue_Id','?')][n]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
for q,d in sorted(perq.items(), key=lambda x:-sum(x[1].values())):
    top=sorted(d.items(), key=lambda x:-x[1])[:6]
    print("queue %s: %.1f ms  "%(q, sum(d.values())/1e3)+", ".join("%s %.2f"%(a.split('<')[0]+('<'+a.split('<')[1] if '<' in a else ''),b/1e3) for a,b in top))
PY
python3 - "$f" <<'PY'
# idle gaps of the busiest queue (the sharded stages' stream): what the GPU waits for between its kernels
import csv,sys,collections,os
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 's3d::' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
l0=[i for i,r in enumerate(rows) if 'k_march_level<2' in r['Kernel_Name']]
last=rows[l0[len(l0)-len(l0)//int(os.environ.get("S3D_NSTEPS","3"))]:]
cnt=collections.Counter(r.get('Queue_Id','?') for r in last)
q=cnt.most_common(1)[0][0]
qs=[r for r in last if r.get('Queue_Id','?')==q]
gaps=[]
for a,b in zip(qs,qs[1:]):
    g=(int(b['Start_Timestamp'])-int(a['End_Timestamp']))/1e3
    if g>40: gaps.append((g,a['Kernel_Name'].split('(')[0].replace('void s3d::','').replace('s3d::',''),b['Kernel_Name'].split('(')[0].replace('void s3d::','').replace('s3d::','')))
print("queue %s: %d gaps > 40 us, %.2f ms in sum"%(q,len(gaps),sum(g for g,_,_ in gaps)/1e3))
agg=collections.defaultdict(lambda:[0,0.0])
for g,a,b in gaps: agg[(a,b)][0]+=1; agg[(a,b)][1]+=g
for (a,b),(n,t) in sorted(agg.items(), key=lambda x:-x[1][1])[:12]: print("  %-28s -> %-28s %3d gaps %8.1f us"%(a,b,n,t))
PY
