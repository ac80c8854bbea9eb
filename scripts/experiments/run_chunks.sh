#!/bin/bash
# variant (c) of the magnitude round trip: chunks whose magnitudes fit the caches (SERHIP_CHUNK_FRAMES)
for cf in 0 112800 56400 28200 14100 7520; do
  echo "== SERHIP_CHUNK_FRAMES=$cf"
  SERHIP_CHUNK_FRAMES=$cf timeout -k 10 120 python bench.py --cpu-clips 0 --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('  ms/step', round(d['ms_per_step'],3), 'stft', round(r['avg_launch_ms']*r['launches_per_step'],3), 'launches', r['launches_per_step'], 'proj', round(r['project_kernel_ms_per_step'],3), 'fin', round(r['finish_kernels_ms_per_step'],3))"
done
