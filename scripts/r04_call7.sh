#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out; mkdir -p $OUT
python3 scripts/step_times_probe.py > $OUT/r04g_step_times.txt 2>&1; grep -v amdgpu.ids $OUT/r04g_step_times.txt
( S3D_DET_EARLY=0 python3 scripts/ab_full.py variants/libsift3d_hip_dev.so | tail -1; S3D_DET_EARLY=1 python3 scripts/ab_full.py variants/libsift3d_hip_dev.so | tail -1; S3D_DET_EARLY=0 python3 scripts/ab_full.py variants/libsift3d_hip_dev.so | tail -1; S3D_DET_EARLY=1 python3 scripts/ab_full.py variants/libsift3d_hip_dev.so | tail -1 ) > $OUT/r04g_det_early.txt 2>&1; grep -v amdgpu.ids $OUT/r04g_det_early.txt
