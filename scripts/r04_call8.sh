#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_pipeline or separable or golden or nondefault or dog_elision or two_stream or async" > $OUT/r04h_pytest.txt 2>&1; echo "pytest rc $?" | tee -a $OUT/r04h_pytest.txt
tail -3 $OUT/r04h_pytest.txt
export S3D_LIB=$PWD/variants/libsift3d_hip_dev.so
for z in 0 50 40 60 30; do S3D_TAG="ZSPLIT=$z" S3D_ZSPLIT=$z python3 scripts/ab_pyramid.py --child 2>&1 | grep pyramid; done > $OUT/r04h_zsplit.txt
cat $OUT/r04h_zsplit.txt
unset S3D_LIB
bash scripts/timeline.sh 512 > $OUT/r04h_timeline.txt 2>&1; cat $OUT/r04h_timeline.txt
