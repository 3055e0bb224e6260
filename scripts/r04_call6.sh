#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out; mkdir -p $OUT
python3 scripts/ab_full.py variants/libsift3d_hip_descold.so variants/libsift3d_hip_qaos_only.so variants/libsift3d_hip_zsym.so > $OUT/r04f_ab_full.txt 2>&1
python3 scripts/ab_full.py variants/libsift3d_hip_descold.so >> $OUT/r04f_ab_full.txt 2>&1
grep -v amdgpu.ids $OUT/r04f_ab_full.txt
python3 scripts/ab_pyramid.py variants/libsift3d_hip_zsym.so > $OUT/r04f_ab_pyramid.txt 2>&1
python3 scripts/ab_pyramid.py variants/libsift3d_hip_zsym.so >> $OUT/r04f_ab_pyramid.txt 2>&1
grep -v amdgpu.ids $OUT/r04f_ab_pyramid.txt
