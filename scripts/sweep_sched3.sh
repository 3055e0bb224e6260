#!/bin/bash
# r04: pyramid stage time at 512^3 over the slot planning / wave priority of octave 1's head against octave 0's widest level, with the
# small octaves in one launch (a -DS3D_DEV_SWITCHES build: scripts/build_variant.sh dev "-DS3D_DEV_SWITCHES" context)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export S3D_LIB=$(realpath variants/libsift3d_hip_dev.so) S3D_AB_NOHASH=1
run() { S3D_TAG="$*" env "$@" python3 scripts/ab_pyramid.py --child 2>&1 | grep pyramid; }
run S3D_PRIO=2
run S3D_SMALL_OCT=0
for prio in 3 2; do for bg in 256 512 768; do for bg1 in 256 512; do run S3D_PRIO=$prio S3D_O0_TAIL_SLOTS=512 S3D_BG1_SLOTS=$bg1 S3D_BG_SLOTS=$bg; done; done; done
run S3D_PRIO=3 S3D_O0_TAIL_SLOTS=640 S3D_BG1_SLOTS=512 S3D_BG_SLOTS=512
run S3D_PRIO=3 S3D_O0_TAIL_SLOTS=512 S3D_BG1_SLOTS=640 S3D_BG_SLOTS=512
run S3D_PRIO=2
