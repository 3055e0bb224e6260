#!/bin/bash
# per-kernel sums of ONE rank's solo step (sift3d_test_sharded_time_rank) of the simulated 8-rank native z-slab run at 1024x1024x512:
#   solo_rank_trace.sh [rank=3]      (FULL=1: every launch in the list, not only the keypoint stages'; ALLK=1: the runtime's own kernels too)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cat > /tmp/solo_one.py <<PY
import importlib, sys
sys.path.insert(0, "$R")
capi = importlib.import_module("3dsift_amd.capi"); synth = importlib.import_module("3dsift_amd.synth")
import torch
vol = synth.blobs_torch((512, 1024, 1024), "cuda", seed=4321).cpu().numpy()
sh = capi.ShardedCSIFT3D(vol, devices=(0,), sim_ranks=8)
for _ in range(3): sh.KpSiftAlgorithm()
r = int(sys.argv[1])
for _ in range(3): t = sh.time_rank(r)
torch.cuda.synchronize()
marker = torch.zeros(7, device="cuda") + 1   # (a foreign kernel marks the start of the last solo run in the trace)
torch.cuda.synchronize()
t = sh.time_rank(r)
print("rank", r, "alone %.3f ms" % (t * 1e3), sh.info()["planes"], sh.info()["stage_partial"])
kp, _ = sh.GetKeypoints(); import numpy as np
print("keypoints per octave", np.bincount(kp["octave"]).tolist(), "per level of octave 0", np.bincount(kp["level"][kp["octave"] == 0]).tolist())
PY
rm -rf /tmp/p_solo; rocprofv3 --kernel-trace -d /tmp/p_solo --output-format csv -- python3 /tmp/solo_one.py ${1:-3} 2>/dev/null | grep rank
f=$(find /tmp/p_solo -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
last_foreign = max(i for i, r in enumerate(rows) if 's3d::' not in r['Kernel_Name'] and 'rocclr' not in r['Kernel_Name'])
import os
last = [r for r in rows[last_foreign + 1:] if 's3d::' in r['Kernel_Name'] or os.environ.get('ALLK')]   # ALLK=1: the runtime's fill / copy kernels too
t0 = int(last[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in last)
tot = collections.defaultdict(float); cnt = collections.Counter()
for r in last:
    n = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('s3d::', '')
    tot[n] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3; cnt[n] += 1
busy = 0; cur = 0
for s, e in sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in last):
    if s > cur: busy += e - s; cur = e
    elif e > cur: busy += e - cur; cur = e
print("solo step: wall %.3f ms, GPU busy %.3f ms, sum of kernel durations %.3f ms, %d launches" % ((t1 - t0) / 1e6, busy / 1e6, sum(tot.values()) / 1e3, len(last)))
for n, v in sorted(tot.items(), key=lambda x: -x[1])[:30]: print("  %-40s %4d launches %9.1f us" % (n, cnt[n], v))
print("in launch order (start offset, duration, queue):")
for r in last:
    n = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('s3d::', '')
    import os
    if os.environ.get('FULL') or 'k_describe' in n or 'k_orient' in n or 'k_mark' in n or 'k_lazy' in n:
        print("   +%8.1f us %8.1f us  q%s grid %s  %s" % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Queue_Id', '?'), r.get('Grid_Size', r.get('Grid_Size_X', '?')), n))
PY
