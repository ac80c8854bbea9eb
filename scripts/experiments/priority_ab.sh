#!/bin/bash
# The issue-priority finding of round 3, reproducible from the shipped tree: builds, OUTSIDE the product tree,
#   ab/noprio.so      every __builtin_amdgcn_s_setprio removed from the kernel headers (the arbiter's default)
#   ab/noprio_fft.so  removed from k_stft_frames only        ab/noprio_proj.so  from k_project_frames only
#   ab/reverse_fft.so k_stft_frames with the grading reversed (young frames first)
# and runs them round robin against the tree's library in ONE call (boxes differ by +-3 %: only same-call numbers
# compare): bench.py (config 3) and config 5 (whisper log-mel).   [gpu]  REPS=3 bash scripts/experiments/priority_ab.sh
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd "$ROOT"
mkdir -p scripts/experiments/ab
build() {   # name, python snippet that edits the copied headers in directory sys.argv[1]
  local name=$1 tmp
  tmp=$(mktemp -d)
  mkdir -p "$tmp/ser_amd"; cp -r ser_amd/csrc "$tmp/ser_amd/csrc"; cp -r include "$tmp/include"
  python3 - "$tmp/ser_amd/csrc" "$name" <<'PY'
import re, sys
from pathlib import Path
root, name = Path(sys.argv[1]), sys.argv[2]
PRIO = re.compile(r"^[ \t]*__builtin_amdgcn_s_setprio\(\d\);.*\n", re.M)
def strip(f):
    p = root / f
    text, n = PRIO.subn("", p.read_text())
    assert n > 0, f
    p.write_text(text)
if name == "noprio":
    for f in ("kernels_fast.hpp", "kernels_project.hpp", "kernels_whisper.hpp"):
        strip(f)
elif name == "noprio_fft":
    strip("kernels_fast.hpp")
elif name == "noprio_proj":
    strip("kernels_project.hpp")
elif name == "reverse_fft":
    p = root / "kernels_fast.hpp"
    text = p.read_text()
    flip = {"0": "3", "1": "2", "2": "1", "3": "0"}
    text, n = re.subn(r"__builtin_amdgcn_s_setprio\((\d)\);", lambda m: f"__builtin_amdgcn_s_setprio({flip[m.group(1)]});", text)
    assert n == 4, n
    p.write_text(text)
PY
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -shared -fPIC -o "scripts/experiments/ab/$name.so" "$tmp/ser_amd/csrc/serhip.hip"
  rm -rf "$tmp"
  echo "built scripts/experiments/ab/$name.so"
}
if [ "${1:-}" = "build" ] || [ ! -f scripts/experiments/ab/noprio.so ]; then
  for v in noprio noprio_fft noprio_proj reverse_fft; do build "$v"; done
  [ "${1:-}" = "build" ] && exit 0
fi
REPS=${REPS:-3} STEPS=${STEPS:-200} bash scripts/experiments/ab_bench.sh
CONFIGS=config5 REPS=${REPS:-3} bash scripts/experiments/ab_configs_libs.sh
