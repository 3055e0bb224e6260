#!/bin/bash
# where the GPU time of the simulated 8-rank z-slab run goes: per-kernel sums of ONE step (rocprofv3 kernel trace of bench.py --workload slab --sim-ranks 8)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/p_ss; rocprofv3 --kernel-trace -d /tmp/p_ss --output-format csv -- python3 $R/bench.py --workload slab --sim-ranks ${1:-8} --steps 1 --warmup 1 > /tmp/p_ss.out 2>/dev/null
f=$(find /tmp/p_ss -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys,collections
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# bench.py runs the workload three times (first call, warm-up, timed step): a step begins with its ranks' level-0 launches
# (k_march_level<2, ...), take everything from the first of the last step's on
s3=[r for r in rows if 's3d::' in r['Kernel_Name']]
l0=[i for i,r in enumerate(s3) if 'k_march_level<2' in r['Kernel_Name']]
nsteps=3
last=s3[l0[len(l0)-len(l0)//nsteps]:]
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
PY
