#!/bin/bash
# Timing-only upper bound for "a fifth wave per SIMD" in k_stft_frames.  A workgroup cannot
# exceed 1024 threads, so five waves per SIMD means TWO workgroups of ten waves per CU.
# late16:        next frame's loads requested at the END of finish_frame (no register overlap with the unpack: the
#                low-register form, 119 VGPRs), exact wait vmcnt(0); 16 waves, results correct.
# two_wg_5waves: the same with 10 waves per workgroup, two workgroups per CU (grid doubled), the transpose buffers
#                ALIASED (s_buf[wave % 6]: 73 KB of LDS per workgroup) and amdgpu_waves_per_eu 5 (96 VGPRs, 15
#                spilled) -- garbage results, but the instruction stream, LDS traffic and occupancy of a kernel
#                that somehow fitted five waves per SIMD.  Time it with EXP_NO_CHROMA=1 (no peak scan: control flow
#                does not depend on the garbage).
# Measured (round 3, 12 500 clips, scripts/experiments/time_kernels.py, round robin in one gpurun call):
#   all flags : shipped 1.92-1.94 ms, late16 1.95-1.96 ms, two_wg_5waves 2.83-2.84 ms
#   no chroma : shipped 1.75-1.77 ms, late16 1.74-1.75 ms, two_wg_5waves 2.56-2.57 ms
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p "$ROOT/scripts/experiments/ab"
for V in late16 two_wg_5waves; do
  TMP=$(mktemp -d)
  mkdir -p "$TMP/ser_amd"; cp -r "$ROOT/ser_amd/csrc" "$TMP/ser_amd/csrc"; cp -r "$ROOT/include" "$TMP/include"
  python3 - "$TMP/ser_amd/csrc" "$V" <<'PY'
import sys
d, v = sys.argv[1], sys.argv[2]
p = d + "/kernels_fast.hpp"
s = open(p).read()
a = "        request_next();\n"
assert s.count(a) == 1
s = s.replace(a, "")
b = "        wave_lds_sync();  // sM reads done before the next frame overwrites the buffer\n"
assert s.count(b) == 1
s = s.replace(b, b + "        request_next();\n")
s = s.replace("wait_samples<kStoreBlock>();", "wait_samples<0>();")
if v == "two_wg_5waves":
    s = s.replace("constexpr int kWaves = 16; ", "constexpr int kWaves = 10; ")
    s = s.replace("v2f s_buf[kWaves][kBufC];", "v2f s_buf[6][kBufC];")
    s = s.replace("v2f* buf = s_buf[wave];", "v2f* buf = s_buf[wave % 6];")
    s = s.replace("__global__ __launch_bounds__(kWaves * 64) void k_stft_frames", "__global__ __launch_bounds__(kWaves * 64, 5) void k_stft_frames")
    q = d + "/serhip.hip"
    t = open(q).read()
    assert "ctx->cu_count = cus;" in t
    open(q, "w").write(t.replace("ctx->cu_count = cus;", "ctx->cu_count = cus * 2;"))
open(p, "w").write(s)
PY
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -shared -fPIC -Rpass-analysis=kernel-resource-usage \
        -o "$ROOT/scripts/experiments/ab/$V.so" "$TMP/ser_amd/csrc/serhip.hip" 2>&1 | grep -A14 "k_stft_framesILb1ELb1" | grep -E "VGPRs|Occupancy|LDS Size|Scratch" || true
  rm -rf "$TMP"
done
