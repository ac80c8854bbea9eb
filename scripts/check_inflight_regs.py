#!/usr/bin/env python3
"""Static check of k_stft_frames' hand-counted waits (ser_amd/csrc/kernels_fast.hpp, "sample loads").

The frame loads are inline asm, so the compiler does not know their destination registers are in flight
until the hand-placed `s_waitcnt vmcnt(N)`.  This script compiles the device code to assembly and verifies,
for every instantiation of the kernel, that
  * no instruction between a frame's load block and the wait reads or writes a destination register
    (no spill, no copy, no reuse), along the loop's textual order: [last in-loop load .. end of function]
    and [loop header .. wait];
  * the kernel has no scratch (spills would be such reads);
  * the store block after the in-loop loads holds exactly the number of vector-memory operations the wait
    assumes.
Run by tests/test_native_host.py (needs hipcc; no GPU).
"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def device_asm() -> str:
    with tempfile.TemporaryDirectory() as tmp:
        out = Path(tmp) / "serhip.s"
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-S",
                        "--cuda-device-only", "-o", str(out), str(ROOT / "ser_amd" / "csrc" / "serhip.hip")],
                       check=True, capture_output=True)
        return out.read_text()


def vregs(text: str) -> set:
    found = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        found |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", text):
        found.add(int(m.group(1)))
    return found


def check_kernel(name: str, body: list) -> list:
    problems = []
    waits = [i for i, l in enumerate(body) if re.search(r"s_waitcnt vmcnt\(\d+\)", l) and
             i > 0 and "ASMSTART" in body[i - 1]]
    loads = [i for i, l in enumerate(body) if "global_load_dwordx2" in l and "offset:" in l]
    if len(waits) != 1 or not loads:
        return [f"{name}: expected one hand-placed wait and asm loads, found {len(waits)} / {len(loads)}"]
    wait = waits[0]
    assumed = int(re.search(r"vmcnt\((\d+)\)", body[wait]).group(1))
    in_loop = [i for i in loads if i > wait]
    dest = set()
    for i in loads:
        dest |= vregs(re.search(r"global_load_dwordx2 (v\[\d+:\d+\])", body[i]).group(1))
    header = max(i for i in range(wait) if "Loop Header" in body[i])
    for i in list(range(header, wait)) + list(range(max(in_loop) + 1, len(body))):
        line = body[i].split(";")[0]
        if "global_load_dwordx2" in line or not line.strip() or line.strip().startswith("."):
            continue
        hit = vregs(line) & dest
        if hit:
            problems.append(f"{name}: line {i}: `{line.strip()}` touches in-flight registers {sorted(hit)}")
    vmem_after = [l for l in body[max(in_loop) + 1:] if re.search(r"\b(global|buffer|flat)_(store|atomic|load)", l)]
    # with peaks on, one more store precedes the block: peaks 64, 65, ... of a frame (rare; it is older than the
    # block, so at worst the wait is longer than needed).  Fewer than assumed would be a bug.
    extra = 1 if name.endswith("Lb1E") else 0
    if len(vmem_after) != assumed + extra:
        problems.append(f"{name}: {len(vmem_after)} vector-memory operations follow the loads, "
                        f"the wait assumes {assumed} (+{extra} conditional)")
    if any("scratch_" in l for l in body):
        problems.append(f"{name}: uses scratch")
    return problems


def main() -> int:
    text = device_asm().split("\n")
    starts = [i for i, l in enumerate(text) if re.match(r"_ZN6serhip13k_stft_framesILb\dELb\dEE.*:", l)]
    problems, seen = [], 0
    for s in starts:
        end = next(i for i in range(s, len(text)) if text[i].startswith(".Lfunc_end"))
        name = re.match(r"_ZN6serhip13k_stft_frames(ILb\dELb\dE)", text[s]).group(1)
        if name == "ILb0ELb0E":
            continue   # no store block, plain vmcnt(0)
        seen += 1
        problems += check_kernel(name, text[s:end])
    if seen != 3:
        problems.append(f"expected 3 instantiations with a store block, found {seen}")
    for p in problems:
        print(p)
    print("check_inflight_regs:", "FAILED" if problems else f"ok ({seen} kernels)")
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
