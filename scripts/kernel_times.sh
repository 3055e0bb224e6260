#!/bin/bash
# per-kernel average durations of one full KpSiftAlgorithm on a 512^3 blob volume (GPU box; rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_kt; rocprofv3 --kernel-trace --stats -d /tmp/p_kt --output-format csv -- python3 ${GRAFT_REPO_ROOT:-/root/repo}/scripts/prof_pyramid.py ${1:-512} 3 5 > /dev/null 2>&1
f=$(find /tmp/p_kt -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if 's3d::' in n:
        n=n.split('(')[0].replace('void ','').replace('s3d::','')
        print(f"{n:28s} calls {int(r['Calls']):4d}  avg {float(r['AverageNs'])/1e3:9.1f} us  max {float(r['MaxNs'])/1e3:9.1f} us  total/3 {float(r['TotalDurationNs'])/3e6:7.3f} ms")
PY
