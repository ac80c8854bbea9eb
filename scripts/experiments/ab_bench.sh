#!/bin/bash
# bench.py with each experimental build of the library (scripts/experiments/ab/*.so) and with the tree's own,
# REPS times round robin (boxes and clocks differ from call to call: only same-call numbers compare)
set -e
REPS=${REPS:-2}
for r in $(seq $REPS); do
for lib in ser_amd/libserhip.so scripts/experiments/ab/*.so; do
  echo "$lib"
  SERHIP_LIBRARY=$lib timeout -k 10 120 python bench.py --cpu-clips 0 --steps ${STEPS:-10} | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('  ms/step', round(d['ms_per_step'],3), 'min', round(d['step_ms_min'],3), 'stft', round(r['avg_launch_ms']*r['launches_per_step'],3), 'proj', round(r['project_kernel_ms_per_step'],3), 'fin', round(r['finish_kernels_ms_per_step'],3))"
done
done
