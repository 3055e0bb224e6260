#!/bin/bash
# r04 GPU call 1: parity suite, A/B of the small-octave launch, small volumes, timeline
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/r04a_pytest.txt 2>&1; echo "pytest rc $?" | tee -a $OUT/r04a_pytest.txt
tail -5 $OUT/r04a_pytest.txt
( python3 scripts/ab_full.py; S3D_SMALL_OCT=0 python3 scripts/ab_full.py variants/libsift3d_hip_dev.so | tail -1 ) > $OUT/r04a_ab_full.txt 2>&1
cat $OUT/r04a_ab_full.txt
( python3 scripts/small_volume_times.py 256 128 64; S3D_SMALL_OCT=0 S3D_LIB=$PWD/variants/libsift3d_hip_dev.so python3 scripts/small_volume_times.py 256 128 64 ) > $OUT/r04a_small_volumes.txt 2>&1
cat $OUT/r04a_small_volumes.txt
bash scripts/timeline.sh 512 > $OUT/r04a_timeline.txt 2>&1
tail -30 $OUT/r04a_timeline.txt
