#!/bin/bash
# pyramid stage time at 512^3 for launch-planning settings (a -DS3D_DEV_SWITCHES build: scripts/build_variant.sh dev "-DS3D_DEV_SWITCHES" context)
export S3D_LIB=$(realpath variants/libsift3d_hip_dev.so) S3D_AB_NOHASH=1
run() { S3D_TAG="$*" env "$@" python3 scripts/ab_pyramid.py --child 2>&1 | grep pyramid; }
run S3D_DEFER_TAIL=0
run S3D_DEFER_TAIL=1
run S3D_DEFER_TAIL=1 S3D_O0_TAIL_SLOTS=768
run S3D_DEFER_TAIL=1 S3D_O0_TAIL_SLOTS=1024
run S3D_DEFER_TAIL=1 S3D_BG_SLOTS=512
run S3D_DEFER_TAIL=1 S3D_BG_SLOTS=768
run S3D_DEFER_TAIL=1 S3D_BG_SLOTS=512 S3D_O0_TAIL_SLOTS=768
run S3D_DEFER_TAIL=1 S3D_BG_SLOTS=768 S3D_O0_TAIL_SLOTS=768
run S3D_DEFER_TAIL=1 S3D_PRIO=1
run S3D_DEFER_TAIL=1 S3D_PRIO=0
