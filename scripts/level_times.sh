#!/bin/bash
# isolated duration of every pyramid launch of one 512^3 run (all octaves on one stream): GPU box, rocprofv3 kernel trace
# usage: scripts/level_times.sh [N=512] [lib.so]
cd /tmp && export TMPDIR=/tmp
export S3D_HOOKS=one_stream=1
[ -n "$2" ] && export S3D_LIB=$(realpath ${GRAFT_REPO_ROOT:-/root/repo}/$2)
rm -rf /tmp/p_lt; rocprofv3 --kernel-trace -d /tmp/p_lt --output-format csv -- python3 ${GRAFT_REPO_ROOT:-/root/repo}/scripts/prof_pyramid.py ${1:-512} 3 1 > /dev/null 2>&1
f=$(find /tmp/p_lt -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys,collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 's3d::' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
per=len(rows)//3
last=rows[-per:]
tot=0
for r in last:
    n=r['Kernel_Name'].split('(')[0].replace('void ','').replace('s3d::','')
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    g=int(r['Grid_Size_X']) if 'Grid_Size_X' in r else 0
    wg=int(r['Workgroup_Size_X']) if 'Workgroup_Size_X' in r else 1
    tot+=d
    if d>15: print(f"{n:26s} grid {g//max(wg,1):6d} wgs  {d:9.1f} us  vgpr {r.get('VGPR_Count','?')} lds {r.get('LDS_Block_Size','?')} scratch {r.get('Scratch_Size', r.get('Private_Segment_Size','?'))}")
print("sum of launches %.3f ms (%d launches)"%(tot/1e3,len(last)))
PY
