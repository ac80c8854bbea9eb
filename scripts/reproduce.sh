#!/bin/bash
# Regenerate every measured artefact of this repository.  Steps marked [gpu] run on an MI355X box
# (inside `gpurun -- '<command>'` on the build pool); the others run anywhere.
#
#   bash scripts/reproduce.sh cpu            # build + CPU test-suite + golden vectors check
#   bash scripts/reproduce.sh gpu <tag>      # [gpu] parity tests, bench line, per-config numbers, rocprofv3 profile
#   bash scripts/reproduce.sh collect <tag>  # copy the summaries merged back under gpurun_out/ into profiles/
set -euo pipefail
cd "$(dirname "$0")/.."
MODE=${1:-cpu}
TAG=${2:-rXX}
case "$MODE" in
  cpu)
    python -c "import __graft_entry__ as g; g.build()"
    python -m pytest tests -x -q -m "not gpu"
    ;;
  gpu)
    mkdir -p gpurun_out
    timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread
    timeout -k 10 300 python bench.py --steps 10 --warmup 2 > "gpurun_out/bench_$TAG.json"
    timeout -k 10 300 python scripts/bench_configs.py > "gpurun_out/configs_$TAG.json"
    bash scripts/profile_gpu.sh "$TAG" > "gpurun_out/prof_$TAG.log" 2>&1
    tail -3 "gpurun_out/prof_$TAG.log"
    ;;
  collect)
    python scripts/collect_profile.py "$TAG"
    [ -f "gpurun_out/configs_$TAG.json" ] && cp "gpurun_out/configs_$TAG.json" "profiles/${TAG}_configs.json"
    ;;
  *) echo "usage: $0 cpu | gpu <tag> | collect <tag>"; exit 2 ;;
esac
