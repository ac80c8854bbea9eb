#!/bin/bash
# bench_configs.py lines (device-resident configs) with the tree's library and every scripts/experiments/ab/*.so, round robin
set -e
REPS=${REPS:-2}
ONLY=${ONLY:-config1,config2_scaled,config4}
for r in $(seq $REPS); do
for lib in ser_amd/libserhip.so scripts/experiments/ab/*.so; do
  echo "$lib"
  SERHIP_LIBRARY=$lib timeout -k 10 200 python scripts/bench_configs.py --only $ONLY 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print('  ', {k: round(v['ms'],4) for k,v in d.items()})"
done
done
