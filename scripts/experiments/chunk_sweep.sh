#!/bin/bash
# bench.py at several scratch chunk sizes (frame slots per chunk): does a chunk whose magnitudes fit the
# 256 MiB Infinity Cache make the projection's reads cache hits?
set -e
for cf in 16384 32768 49152 65536 131072 2097152; do
  echo "chunk_frames $cf"
  SERHIP_CHUNK_FRAMES=$cf timeout -k 10 120 python bench.py --cpu-clips 0 --steps 10 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('  ms/step', round(d['ms_per_step'],3), 'stft', round(r['avg_launch_ms']*r['launches_per_step'],3), 'proj', round(r['project_kernel_ms_per_step'],3), 'fin', round(r['finish_kernels_ms_per_step'],3), 'launches', r['launches_per_step'])"
done
