#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out; mkdir -p $OUT
export S3D_LIB=$PWD/variants/libsift3d_hip_dev.so
for c in 1 0 1 0 1 0; do echo "CHAIN=$c $(S3D_CHAIN=$c python3 scripts/ab_full.py --child 2>&1 | grep total)"; done > $OUT/r04q_chain.txt 2>&1
for c in 1 0 1 0; do S3D_CHAIN=$c python3 scripts/small_volume_times.py 256 128 64 2>&1 | grep -v amdgpu | sed "s/^/CHAIN=$c /"; done >> $OUT/r04q_chain.txt 2>&1
cat $OUT/r04q_chain.txt
