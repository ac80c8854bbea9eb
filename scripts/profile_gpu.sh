#!/bin/bash
# Profile the bench workload on the GPU box: kernel trace + stats, then PMC passes (each its
# own run; counters never combined with trace domains other than --kernel-trace).
# Usage (inside gpurun): bash scripts/profile_gpu.sh <tag>
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# 55 launches per kernel: the first ones (cold scratch pages, cold tables, clocks still ramping: +35 % on the first
# k_project_frames, +5 % on the first ten steps as a whole) weigh little in the averages
BENCH="python3 $ROOT/bench.py --steps 50 --warmup 5 --cpu-clips 0"
# the plain run first (no profiler): the bench line with the CPU baseline
( cd "$ROOT" && timeout -k 10 400 python3 bench.py > "$ROOT/gpurun_out/bench_$TAG.json" 2> "$ROOT/gpurun_out/bench_$TAG.err" ) || { echo "bench failed"; tail -5 "$ROOT/gpurun_out/bench_$TAG.err"; exit 1; }
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
# the bench line printed INSIDE the traced run: HIP-event kernel times under the profiler, to set beside the
# profiler's own durations of the same launches (is the difference the clock of the two methods, or the kernels?)
grep -a '^{"metric"' "$OUT/trace.log" > "$OUT/bench_in_trace_$TAG.json" || true
echo "trace done"
i=0
for CTRS in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d "$OUT/pmc$i" -- $BENCH > "$OUT/pmc$i.log" 2>&1 || { echo "pmc pass $i ($CTRS) failed"; tail -3 "$OUT/pmc$i.log"; }
  echo "pmc pass $i done"
done
python3 $ROOT/scripts/summarize_profile.py "$OUT" "$TAG" > "$OUT/summary.txt" 2>&1
# keep only the small artefacts (gpurun copies back at most 64 MiB)
find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_$TAG.csv" \;
rm -rf "$OUT"/trace "$OUT"/pmc[0-9]*
grep -E "k_stft_frames|k_segment_finish|k_project_frames" -A 26 "$OUT/summary.txt" | head -120
