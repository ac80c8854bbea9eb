#!/bin/bash
# scripts/bench_configs.py --only <configs> with the tree's library and with every scripts/experiments/ab/*.so,
# REPS times round robin in one call:  CONFIGS=config5 REPS=3 bash scripts/experiments/ab_configs_libs.sh
set -e
for r in $(seq ${REPS:-2}); do
for lib in ser_amd/libserhip.so scripts/experiments/ab/*.so; do
  SERHIP_LIBRARY=$lib timeout -k 10 200 python scripts/bench_configs.py --only ${CONFIGS:-config5} 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print('$lib', {k: round(v.get('ms', 0), 4) for k, v in d.items() if isinstance(v, dict)})"
done
done
