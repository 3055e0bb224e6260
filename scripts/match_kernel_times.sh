#!/bin/bash
# per-kernel durations of the matcher (scripts/bench_match.py under rocprofv3 --kernel-trace --stats), default library + variants:
#   bash scripts/match_kernel_times.sh [variants/lib...so ...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for lib in default "$@"; do
  if [ "$lib" = default ]; then unset S3D_LIB; else export S3D_LIB=$(realpath $R/$lib); fi
  rm -rf /tmp/p_mk; rocprofv3 --kernel-trace --stats -d /tmp/p_mk --output-format csv -- python3 $R/scripts/bench_match.py > /tmp/p_mk.out 2>&1
  echo "== $lib"; grep -E "injectMatch|enhancedMatch" /tmp/p_mk.out
  f=$(find /tmp/p_mk -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if any(k in n for k in ('k_scores','k_merge','k_rescore','k_exact','k_row_norm')):
        print("  %-60s calls %4s  avg %9.1f us  min %9.1f  max %9.1f"%(n.split('(')[0][-60:], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
  t=$(find /tmp/p_mk -name "*kernel_trace.csv" | head -1)
  python3 - "$t" <<'PY'
import csv,sys
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(sys.argv[1])) if 'k_scores' in r['Kernel_Name']]
print("  k_scores launches (us):", " ".join("%.0f"%x for x in d))
PY
done
