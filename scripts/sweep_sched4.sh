#!/bin/bash
# r04 (64 x 32 tiles for octave 0's levels up to the seed level, ring-less small octaves): the slot planning of the overlap region again
# (a -DS3D_DEV_SWITCHES build: scripts/build_variant.sh dev "-DS3D_DEV_SWITCHES" entry_test)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export S3D_LIB=$(realpath variants/libsift3d_hip_dev.so) S3D_AB_NOHASH=1
run() { S3D_TAG="$*" env "$@" python3 scripts/ab_pyramid.py --child 2>&1 | grep pyramid; }
run S3D_PRIO=3
for tail in 384 512 640 768; do for bg1 in 256 384 512; do run S3D_O0_TAIL_SLOTS=$tail S3D_BG1_SLOTS=$bg1; done; done
run S3D_BG_SLOTS=512
run S3D_BG_SLOTS=128
run S3D_DEFER_TAIL=1 S3D_BG1_SLOTS=768
run S3D_DEFER_TAIL=1 S3D_BG1_SLOTS=512
run S3D_DEFER_TAIL=1 S3D_BG1_SLOTS=768 S3D_O0_TAIL_SLOTS=768
run S3D_PRIO=2
run S3D_PRIO=3
