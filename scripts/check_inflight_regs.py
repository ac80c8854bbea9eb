#!/usr/bin/env python3
"""Static proof obligations of k_stft_frames' hand-counted waits (ser_amd/csrc/kernels_fast.hpp, "sample loads").

The frame loads are inline asm, so the compiler does not know that their destination registers are in flight
until the hand-placed `s_waitcnt vmcnt(N)`.  This script compiles the device code to assembly, rebuilds the
control-flow graph of every instantiation of the kernel and runs a forward data-flow analysis over it:

  state = (F, c):  F = registers that MAY be in flight (union over paths),
                   c = the FEWEST vector-memory operations issued since the last asm load (min over paths)
  asm load (global_load_dwordx2 inside #ASMSTART)   F |= its destination pair, c = 0
  any other vector-memory operation                 c += 1
  s_waitcnt vmcnt(0)                                F = {}      (everything has landed)
  hand-placed s_waitcnt vmcnt(N), N > 0             requires c >= N on every path (else a load could still be
                                                    among the N operations left outstanding), then F = {}
  any other instruction that names a register in F  ERROR: read, write, copy or spill of a register in flight

The compiler linearises `if (a) {..} else if (b) {..}` into blocks that re-test a flag it has just set
(`s_mov_b64 s[x:y], 0 | -1` ... `s_andn2_b64 vcc, exec, s[x:y]` + `s_cbranch_vccnz`): taken literally, a path
would run the 16-pair load block and then the 4-pair one.  The analysis therefore propagates those 64-bit 0 / -1
constants (nothing else) and follows only the feasible side of a branch whose condition they decide.

and checks that no instantiation uses scratch.  Run by tests/test_native_host.py (needs hipcc; no GPU).
"""
import os
import re
import shlex
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
BRANCH = re.compile(r"^\s*(s_branch|s_cbranch_\w+)\s+(\.LBB\d+_\d+)")
LABEL = re.compile(r"^(\.LBB\d+_\d+):")
VMEM = re.compile(r"^\s*(global|buffer|flat|scratch)_(load|store|atomic)\w*")


def device_asm() -> str:
    with tempfile.TemporaryDirectory() as tmp:
        out = Path(tmp) / "serhip.s"
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-S",
                        "--cuda-device-only", *shlex.split(os.environ.get("SERHIP_CHECK_FLAGS", "")),   # e.g. -DSERHIP_EXP_...=1 with SERHIP_CHECK_SOURCE=<patched copy>/serhip.hip
                        "-o", str(out), os.environ.get("SERHIP_CHECK_SOURCE", str(ROOT / "ser_amd" / "csrc" / "serhip.hip"))],
                       check=True, capture_output=True)
        return out.read_text()


def vregs(text: str) -> frozenset:
    found = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        found |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", text):
        found.add(int(m.group(1)))
    return frozenset(found)


def parse(body):
    """-> list of instructions (text, in_asm) and blocks [(first, last_exclusive, successors)]"""
    insts, label_at, in_asm = [], {}, False
    for raw in body:
        if "#ASMSTART" in raw:
            in_asm = True
            continue
        if "#ASMEND" in raw:
            in_asm = False
            continue
        m = LABEL.match(raw)
        if m:
            label_at[m.group(1)] = len(insts)
            continue
        line = raw.split(";")[0].strip()
        if not line or line.startswith("."):
            continue
        insts.append((line, in_asm))
    leaders = {0} | set(label_at.values())
    for i, (line, _) in enumerate(insts):
        if BRANCH.match(line) or line.startswith("s_endpgm"):
            leaders.add(i + 1)
    starts = sorted(x for x in leaders if x < len(insts))
    blocks = []
    for k, first in enumerate(starts):
        last = starts[k + 1] if k + 1 < len(starts) else len(insts)
        tail = insts[last - 1][0]
        succ = []
        m = BRANCH.match(tail)
        if m:
            succ.append(label_at[m.group(2)])
            if m.group(1) != "s_branch" and last < len(insts):
                succ.append(last)
        elif not tail.startswith("s_endpgm") and last < len(insts):
            succ.append(last)
        blocks.append((first, last, succ))
    return insts, blocks


def check_kernel(name: str, body: list) -> list:
    problems = []
    insts, blocks = parse(body)
    if any(line.startswith("scratch_") for line, _ in insts):
        problems.append(f"{name}: uses scratch (a spill of a register in flight would go unnoticed by the hardware)")
    hand_waits = [i for i, (line, a) in enumerate(insts) if a and line.startswith("s_waitcnt vmcnt(")]
    asm_loads = [i for i, (line, a) in enumerate(insts) if a and line.startswith("global_load_dwordx2")]
    if len(hand_waits) != 1 or len(asm_loads) < 30:
        problems.append(f"{name}: expected one hand-placed wait and the asm load blocks, found {len(hand_waits)} / {len(asm_loads)}")
        return problems
    by_start = {b[0]: b for b in blocks}
    # states are kept apart per flag environment (trace partitioning): merging the environments at the join in
    # front of a re-test would forget exactly the constant that decides it
    state_in = {(0, frozenset()): (frozenset(), 10 ** 6)}
    work = [(0, frozenset())]
    reported = set()

    def sregs(tok):
        m = re.match(r"s\[(\d+):(\d+)\]$", tok)
        if m:
            return set(range(int(m.group(1)), int(m.group(2)) + 1))
        m = re.match(r"s(\d+)$", tok)
        return {int(m.group(1))} if m else set()

    while work:
        first, consts_f = work.pop()
        _, last, succ = by_start[first]
        flight, count = state_in[(first, consts_f)]
        consts = dict(consts_f)          # "s[a:b]" or "vcc" -> 0 (all lanes off) | 1 (== exec, nonzero)
        for i in range(first, last):
            line, in_asm = insts[i]
            # ---- 64-bit flag constants
            ops = [t.strip() for t in line.split(None, 1)[1].split(",")] if " " in line else []
            if line.startswith("s_") and ops and not line.startswith(("s_cbranch", "s_branch", "s_waitcnt", "s_nop")):
                dst = ops[0]
                value = None
                if line.startswith("s_mov_b64") and len(ops) == 2 and ops[1] in ("0", "-1"):
                    value = 0 if ops[1] == "0" else 1
                elif line.startswith("s_andn2_b64") and len(ops) == 3 and ops[1] == "exec" and ops[2] in consts:
                    value = 1 - consts[ops[2]]
                elif line.startswith("s_and_b64") and len(ops) == 3 and ops[1] == "exec" and ops[2] in consts:
                    value = consts[ops[2]]
                elif line.startswith("s_mov_b64") and len(ops) == 2 and ops[1] in consts:
                    value = consts[ops[1]]
                written = sregs(dst)
                for key in list(consts):
                    if key == dst or (written and written & sregs(key)):
                        del consts[key]
                if line.startswith("s_cmp") or line.startswith("s_bitcmp"):
                    pass                  # writes scc only
                elif value is not None:
                    consts[dst] = value
                if dst != "vcc" and any(line.startswith(x) for x in ("s_and_saveexec", "s_or_saveexec")):
                    consts.pop("vcc", None)
            elif line.startswith("v_cmp") or "vcc" in line.split(None, 1)[-1].split(",")[0:1]:
                consts.pop("vcc", None)
            if in_asm and line.startswith("global_load_dwordx2"):
                flight = flight | vregs(line.split(",")[0])
                count = 0
                continue
            m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", line)
            if m:
                need = int(m.group(1))
                if need == 0:
                    flight = frozenset()
                elif in_asm:
                    if count < need and ("count", i) not in reported:
                        reported.add(("count", i))
                        problems.append(f"{name}: only {count} vector-memory operations are sure to follow the loads "
                                        f"on some path, the hand-placed wait assumes {need}")
                    flight = frozenset()
                continue
            if VMEM.match(line):
                count += 1
            hit = vregs(line) & flight
            if hit and ("touch", i) not in reported:
                reported.add(("touch", i))
                problems.append(f"{name}: instruction {i} `{line}` touches registers in flight {sorted(hit)}")
        feasible = list(succ)
        tail = insts[last - 1][0]
        m = BRANCH.match(tail)
        if m and m.group(1) in ("s_cbranch_vccnz", "s_cbranch_vccz") and "vcc" in consts and len(succ) == 2:
            taken = (consts["vcc"] == 1) == (m.group(1) == "s_cbranch_vccnz")
            feasible = [succ[0]] if taken else [succ[1]]
        env = frozenset(consts.items())
        for nxt in feasible:
            old = state_in.get((nxt, env))
            new = (flight, count) if old is None else (old[0] | flight, min(old[1], count))
            if new != old:
                state_in[(nxt, env)] = new
                work.append((nxt, env))
    return problems


def main() -> int:
    text = device_asm().split("\n")
    starts = [i for i, l in enumerate(text) if re.match(r"_ZN6serhip13k_stft_framesILb\dELb\dEE.*:", l)]
    problems, seen = [], 0
    for s in starts:
        end = next(i for i in range(s, len(text)) if text[i].startswith(".Lfunc_end"))
        name = re.match(r"_ZN6serhip13k_stft_frames(ILb\dELb\dE)", text[s]).group(1)
        if name == "ILb0ELb0E":
            continue   # no store block: its wait is a plain vmcnt(0)
        seen += 1
        problems += check_kernel(name, text[s + 1:end])
    if seen != 3:
        problems.append(f"expected 3 instantiations with a store block, found {seen}")
    for p in problems:
        print(p)
    print("check_inflight_regs:", "FAILED" if problems else f"ok ({seen} kernels)")
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
