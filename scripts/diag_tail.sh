T="tests/test_gpu_sharded_native.py tests/test_gpu_slab.py"
run() { echo "== $*"; env "$@" timeout -k 10 200 python3 -m pytest $T -m gpu -q 2>&1 | grep "^FAILED\|passed\|failed"; }
run A=1
export S3D_LIB=$PWD/variants/libsift3d_hip_dev.so
run S3D_CHAIN=0
run S3D_SMALL_OCT=0
run SIFT3D_HOOK_DESC_NOSPLIT=1
run S3D_DET_EARLY=0
