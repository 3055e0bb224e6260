#!/bin/bash
# what happens inside the largest idle gap of the simulated native z-slab step: every kernel (any queue, HIP's own fill / copy kernels included),
# every memory copy and the host's HIP calls that overlap it  (rocprofv3 kernel + memory-copy + HIP runtime trace, no counters)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/p_gap; rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace -d /tmp/p_gap --output-format csv -- python3 $R/bench.py --workload slab --native --sim-ranks ${1:-8} --steps 1 --warmup 1 > /tmp/p_gap.out 2>/dev/null
python3 - /tmp/p_gap <<'PY'
import csv, sys, glob, collections
d = sys.argv[1]
def load(pat):
    f = glob.glob(d + "/**/*" + pat, recursive=True)
    return [r for r in csv.DictReader(open(f[0]))] if f else []
K = load("kernel_trace.csv"); M = load("memory_copy_trace.csv"); A = load("hip_api_trace.csv")
K.sort(key=lambda r: int(r['Start_Timestamp']))
s3 = [r for r in K if 's3d::' in r['Kernel_Name']]
l0 = [i for i, r in enumerate(s3) if 'k_march_level<2' in r['Kernel_Name']]
last = s3[l0[len(l0) - len(l0) // 2]:]
q = collections.Counter(r['Queue_Id'] for r in last).most_common(1)[0][0]
qs = [r for r in last if r['Queue_Id'] == q]
gaps = sorted(((int(b['Start_Timestamp']) - int(a['End_Timestamp']), a, b) for a, b in zip(qs, qs[1:])), key=lambda x: -x[0])
nm = lambda r: r['Kernel_Name'].split('(')[0].replace('void ', '').replace('s3d::', '')[:60]
for g, a, b in gaps[:3]:
    t0, t1 = int(a['End_Timestamp']), int(b['Start_Timestamp'])
    print("== gap %.1f us between %s and %s" % (g / 1e3, nm(a), nm(b)))
    ev = []
    for r in K:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if e > t0 and s < t1 and r is not a and r is not b: ev.append((s, "kernel q%s %-50s %.1f us" % (r['Queue_Id'], nm(r), (e - s) / 1e3)))
    for r in M:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if e > t0 and s < t1: ev.append((s, "copy   %s %s bytes %.1f us" % (r.get('Direction', '?'), r.get('Bytes', r.get('Size', '?')), (e - s) / 1e3)))
    for r in A:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if e > t0 and s < t1 and (e - s) > 20000: ev.append((s, "host   %-40s %.1f us" % (r['Function'], (e - s) / 1e3)))
    ev.sort()
    for s, txt in ev[:60]: print("   +%8.1f us  %s" % ((s - t0) / 1e3, txt))
    if len(ev) > 60: print("   ... %d more" % (len(ev) - 60))
PY
