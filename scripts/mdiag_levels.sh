#!/bin/bash
# isolated level times of the timing-only builds of k_march_level (S3D_MDIAG bits), 512^3:  scripts/mdiag_levels.sh md16 md32 ...
for v in "" "$@"; do
  echo "== ${v:-product}"
  if [ -z "$v" ]; then bash scripts/level_times.sh 512 2>/dev/null | grep "768 wgs\|512 wgs"; else bash scripts/level_times.sh 512 variants/libsift3d_hip_$v.so 2>/dev/null | grep "768 wgs\|512 wgs"; fi
done
