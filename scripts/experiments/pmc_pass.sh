#!/bin/bash
# one rocprofv3 PMC pass over the bench workload: bash scripts/experiments/pmc_pass.sh <tag> <counters...>
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/pmc1" -- python3 $ROOT/bench.py --steps 3 --warmup 1 --cpu-clips 0 > "$OUT/pmc.log" 2>&1 || { echo "pmc pass failed"; tail -5 "$OUT/pmc.log"; exit 1; }
python3 $ROOT/scripts/summarize_profile.py "$OUT" "$TAG" | grep -E "^k_|^    " | grep -B0 -A40 "k_project_frames" | head -45
rm -rf "$OUT/pmc1"
