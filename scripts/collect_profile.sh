#!/bin/bash
# Collect the per-round rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   bash scripts/collect_profile.sh r01c
# writes gpurun_out/<tag>_*; copy the files into profiles/ afterwards.
set -e
TAG=${1:-rXX}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python3 $REPO/bench.py --steps 5 --warmup 2 --cpu-sample 0 > $OUT/${TAG}_bench_under_rocprof.json 2> /tmp/prof_$TAG.err || true
f=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1)
cp "$f" $OUT/${TAG}_rocprofv3_kernel_stats.csv
python3 $REPO/bench.py > $OUT/${TAG}_bench.json 2> /dev/null
python3 $REPO/scripts/measure_traffic.py 512 > $OUT/${TAG}_pyramid_traffic_512.json
head -12 $OUT/${TAG}_rocprofv3_kernel_stats.csv
