#!/bin/bash
# rocprofv3 evidence for everything that is NOT the bench command: one kernel trace and three counter passes per
# configuration of scripts/bench_configs.py, each configuration in runs of its own so that kernels shared between
# configurations (k_stft_frames, ...) are accounted per configuration.
# Usage (inside gpurun): bash scripts/profile_configs.sh <tag> [config ...]
set -u
TAG=${1:-r03}; shift || true
CONFIGS=${*:-"config1 config2_scaled config4 config5 generic_short classifier pooling prepare"}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_configs_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for CFG in $CONFIGS; do
  D=$OUT/$CFG
  mkdir -p "$D"
  RUN="python3 $ROOT/scripts/bench_configs.py --only $CFG"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/trace" -- $RUN --out "$D/result.json" > "$D/trace.log" 2>&1 || { echo "$CFG: trace failed"; tail -3 "$D/trace.log"; exit 1; }
  i=0
  for CTRS in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES SQ_WAVE_CYCLES"; do
    i=$((i+1))
    timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d "$D/pmc$i" -- $RUN > /dev/null 2> "$D/pmc$i.log" || { echo "$CFG: pmc pass $i failed"; tail -3 "$D/pmc$i.log"; exit 1; }
  done
  python3 $ROOT/scripts/summarize_profile.py "$D" "$CFG" > "$D/summary.txt" 2>&1
  rm -rf "$D"/trace "$D"/pmc[0-9]
  echo "$CFG profiled"
done
python3 $ROOT/scripts/configs_table.py "$OUT" "$TAG"
