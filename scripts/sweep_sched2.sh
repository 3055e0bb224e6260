#!/bin/bash
# pyramid stage time at 512^3 for the slot planning of octave 0's tail / the small octaves' chains and the wave priority of octave 1
# (a -DS3D_DEV_SWITCHES build: scripts/build_variant.sh dev "-DS3D_DEV_SWITCHES" context)
export S3D_LIB=$(realpath variants/libsift3d_hip_dev.so) S3D_AB_NOHASH=1
run() { S3D_TAG="$*" env "$@" python3 scripts/ab_pyramid.py --child 2>&1 | grep pyramid; }
for prio in 2 1; do for tail in 384 512 768; do for bg in 256 384 512; do run S3D_PRIO=$prio S3D_O0_TAIL_SLOTS=$tail S3D_BG_SLOTS=$bg; done; done; done
