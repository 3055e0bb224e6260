#!/bin/bash
# start/end (us, relative to the first launch) of every kernel of the last full KpSiftAlgorithm (stage 5, blob volume), pyramid kernels omitted
#   bash scripts/timeline_full.sh [N=512] [lib.so]
cd /tmp && export TMPDIR=/tmp
[ -n "$2" ] && export S3D_LIB=$(realpath ${GRAFT_REPO_ROOT:-/root/repo}/$2)
rm -rf /tmp/p_tf; rocprofv3 --kernel-trace -d /tmp/p_tf --output-format csv -- python3 ${GRAFT_REPO_ROOT:-/root/repo}/scripts/prof_pyramid.py ${1:-512} 3 5 > /dev/null 2>&1
f=$(find /tmp/p_tf -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 's3d::' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the last run starts at the last k_march_level<2 launch
starts=[i for i,r in enumerate(rows) if 'k_march_level<2' in r['Kernel_Name']]
last=rows[starts[-1]:]
t0=int(last[0]['Start_Timestamp'])
for r in last:
    n=r['Kernel_Name'].split('(')[0].replace('void ','').replace('s3d::','')
    if n.startswith('k_march') or n.startswith('k_conv') or n.startswith('k_downsample'): continue
    s=(int(r['Start_Timestamp'])-t0)/1e3; e=(int(r['End_Timestamp'])-t0)/1e3
    g=int(r['Grid_Size_X'])//max(int(r['Workgroup_Size_X']),1)
    print(f"{n:22s} wgs {g:6d} q {r.get('Queue_Id','?'):>3s}  {s:8.1f} -> {e:8.1f}  ({e-s:7.1f} us)")
PY
