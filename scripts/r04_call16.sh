#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $OUT/r04p_pytest.txt 2>&1; echo "pytest rc $?" | tee -a $OUT/r04p_pytest.txt
tail -3 $OUT/r04p_pytest.txt
export S3D_LIB=$PWD/variants/libsift3d_hip_dev.so
( for c in 1 0 1 0; do S3D_CHAIN=$c python3 scripts/ab_full.py --child 2>&1 | grep total | sed "s/^/CHAIN=$c /"; done
  for c in 1 0; do S3D_CHAIN=$c python3 scripts/small_volume_times.py 256 128 64 2>&1 | grep -v amdgpu | sed "s/^/CHAIN=$c /"; done ) > $OUT/r04p_chain.txt
cat $OUT/r04p_chain.txt
unset S3D_LIB
bash scripts/timeline.sh 512 > $OUT/r04p_timeline.txt 2>&1; tail -22 $OUT/r04p_timeline.txt
