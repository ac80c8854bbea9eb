#!/bin/bash
# kernel times of timing-only builds (garbage outputs) and of the chroma-only flag set
for lib in ser_amd/libserhip.so scripts/experiments/stamps/*.so; do
  echo "== $lib"
  SERHIP_LIBRARY=$lib timeout -k 10 120 python scripts/experiments/time_kernels.py 2>&1 | grep -v amdgpu.ids | tail -1
done
echo "== chroma only (ser_amd/libserhip.so)"
EXP_CHROMA_ONLY=1 timeout -k 10 120 python scripts/experiments/time_kernels.py 2>&1 | grep -v amdgpu.ids | tail -1
