#!/bin/bash
# Regenerate every measured artefact of this repository.  Steps marked [gpu] run on an MI355X box
# (inside `gpurun -- '<command>'` on the build pool); the others run anywhere.
#
#   bash scripts/reproduce.sh cpu            # build + CPU test-suite + golden vectors check
#   bash scripts/reproduce.sh gpu <tag>      # [gpu] parity tests; bench line + rocprofv3 trace + PMC passes of the bench
#                                            #       command; per-configuration traces + counters; end-to-end ingest lines;
#                                            #       copy/kernel overlap trace of the ingest pipelines; 2-rank rehearsal
#   bash scripts/reproduce.sh parity <tag>   # [gpu] parity at scale against the oracle (3 000 / 1 500 clips, 3 600 windows)
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
    bash scripts/profile_gpu.sh "$TAG" > "gpurun_out/prof_$TAG.log" 2>&1 || { tail -5 "gpurun_out/prof_$TAG.log"; exit 1; }
    grep -E "^k_" "gpurun_out/prof_$TAG/summary.txt"
    bash scripts/profile_configs.sh "$TAG" > "gpurun_out/prof_configs_$TAG.log" 2>&1 || { tail -5 "gpurun_out/prof_configs_$TAG.log"; exit 1; }
    tail -12 "gpurun_out/prof_configs_$TAG.log"
    timeout -k 10 400 python scripts/bench_configs.py --only config4_host_api,config4_from_file,config2_from_files,config2_from_files_x4,config3_from_host,config3_backend_api \
        --out "gpurun_out/ingest_$TAG.json" 2> "gpurun_out/ingest_$TAG.err"
    ( cd /tmp && TMPDIR=/tmp timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv \
        -d "$OLDPWD/gpurun_out/ingest_trace_$TAG" -- python3 "$OLDPWD/scripts/ingest_trace.py" > "$OLDPWD/gpurun_out/ingest_trace_$TAG.log" 2>&1 )
    python3 scripts/overlap_report.py "gpurun_out/ingest_trace_$TAG" > "gpurun_out/ingest_overlap_$TAG.json"
    rm -rf "gpurun_out/ingest_trace_$TAG"
    timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
        bench.py --gpus 2 --steps 3 --warmup 1 --clips-per-gpu 3000 --rehearse-on-one-gpu 2> /dev/null | grep -a '^{"metric"' > "gpurun_out/rehearsal_$TAG.json"
    ;;
  parity)
    mkdir -p gpurun_out
    timeout -k 10 900 python scripts/parity_at_scale.py --clips 3000 --kind harmonic --out "gpurun_out/parity_harmonic_3000_$TAG.json"
    timeout -k 10 900 python scripts/parity_at_scale.py --clips 1500 --kind noisy --out "gpurun_out/parity_noisy_1500_$TAG.json"
    timeout -k 10 900 python scripts/parity_at_scale.py --clips 3600 --kind windows --out "gpurun_out/parity_windows_3600_$TAG.json"
    ;;
  collect)
    python scripts/collect_profile.py "$TAG"
    for f in "configs_$TAG.json:${TAG}_configs.json" "configs_$TAG.md:${TAG}_configs.md"; do
      [ -f "gpurun_out/prof_configs_$TAG/${f%%:*}" ] && cp "gpurun_out/prof_configs_$TAG/${f%%:*}" "profiles/${f##*:}"
    done
    for cfg in gpurun_out/prof_configs_$TAG/*/; do
      name=$(basename "$cfg")
      [ -f "$cfg/summary_$name.json" ] && cp "$cfg/summary_$name.json" "profiles/${TAG}_configs_${name}_summary.json"
    done
    [ -f "gpurun_out/ingest_$TAG.json" ] && cp "gpurun_out/ingest_$TAG.json" "profiles/${TAG}_ingest.json"
    [ -f "gpurun_out/ingest_overlap_$TAG.json" ] && cp "gpurun_out/ingest_overlap_$TAG.json" "profiles/${TAG}_ingest_overlap.json"
    [ -f "gpurun_out/rehearsal_$TAG.json" ] && cp "gpurun_out/rehearsal_$TAG.json" "profiles/${TAG}_rehearsal_2ranks_one_gpu.json"
    for k in harmonic_3000 noisy_1500 windows_3600; do
      if [ -f "gpurun_out/parity_${k}_$TAG.json" ]; then cp "gpurun_out/parity_${k}_$TAG.json" "profiles/${TAG}_parity_${k}.json"; fi
    done
    ;;
  *) echo "usage: $0 cpu | gpu <tag> | parity <tag> | collect <tag>"; exit 2 ;;
esac
