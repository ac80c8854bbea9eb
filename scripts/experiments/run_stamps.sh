#!/bin/bash
# per-phase stamps of the diagnostic builds, then plain kernel times of the timed builds
for lib in "$@"; do
  echo "== $lib"
  SERHIP_LIBRARY=$lib timeout -k 10 120 python scripts/experiments/time_kernels.py 2>&1 | grep -v amdgpu.ids | tail -2
done
