#!/bin/bash
# isolated level times of the timing-only builds of k_march_level (S3D_MDIAG bits), 512^3
for v in "" md16 md32 md64 md112 md8 md4; do
  echo "== ${v:-product}"
  if [ -z "$v" ]; then bash scripts/level_times.sh 512 | head -8 | grep "768 wgs\|512 wgs"; else bash scripts/level_times.sh 512 variants/libsift3d_hip_$v.so | head -8 | grep "768 wgs\|512 wgs"; fi
done
