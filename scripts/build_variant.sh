#!/bin/bash
# build a kernel-variant library for A/B timing:
#   scripts/build_variant.sh NAME "-DS3D_...=..." [source=kernels_march]   ->  variants/libsift3d_hip_NAME.so
set -e
cd "$(dirname "$0")/.."
name=$1; extra=$2; src=${3:-kernels_march}
# (the development switches -- -DS3D_DEV_SWITCHES: environment overrides of tuning values and hooks -- live in entry_test.hip)
B=build/variants/$name   # under csrc/build: git-ignored and .gpurunignore'd (only the linked .so under variants/ travels)
mkdir -p variants 3dsift_amd/csrc/$B
cd 3dsift_amd/csrc
for f in context tables entry_free entry_slab entry_test staging sharded kernels_pyramid kernels_march kernels_small kernels_detect kernels_orient kernels_desc kernels_match; do
  [ "$f" != "$src" ] && [ -f build/$f.o ] && cp -u build/$f.o $B/$f.o
done
FL="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -Wno-unused-value"
{ [ "$src" = kernels_march ] || [ "$src" = kernels_small ] || [ "$src" = kernels_desc ] || [ "$src" = kernels_orient ]; } && FL="$FL -fno-slp-vectorize"
/opt/rocm/bin/hipcc $extra $FL -c $src.hip -o $B/$src.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/libsift3d_hip_$name.so $B/*.o
echo built variants/libsift3d_hip_$name.so
