cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_kt2; rocprofv3 --kernel-trace -d /tmp/p_kt2 --output-format csv -- python3 ${GRAFT_REPO_ROOT:-/root/repo}/scripts/prof_pyramid.py 512 2 5 > /dev/null 2>&1
f=$(find /tmp/p_kt2 -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'k_lazy' in r['Kernel_Name'] or 'k_mark' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
for r in rows[-14:]:
    n=r['Kernel_Name'].split('(')[0].replace('void ','').replace('s3d::','')
    print(n, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, 'us grid', r.get('Grid_Size_X'))
PY
