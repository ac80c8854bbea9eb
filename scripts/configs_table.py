#!/usr/bin/env python3
"""Per-configuration roofline table from scripts/profile_configs.sh output.

Usage: configs_table.py <gpurun_out/prof_configs_TAG> <TAG>
Writes <dir>/configs_<TAG>.json and .md: per configuration the wall-clock figure of the un-profiled loop inside the
traced run, the library's kernels with calls, average duration (rocprofv3 kernel trace), HBM bytes (2 x FETCH_SIZE +
WRITE_SIZE, the guide's gfx950 correction) and VALU / LDS / MFMA instruction counts, and the fractions of the 8 TB/s
HBM roofline: algorithmic bytes / wall time and algorithmic bytes / summed kernel time per invocation.  Everything in
the table can be recomputed from the JSON copied to profiles/."""
import json
import sys
from pathlib import Path

HBM = 8.0e12


def main():
    root, tag = Path(sys.argv[1]), sys.argv[2]
    table = {}
    for d in sorted(p for p in root.iterdir() if p.is_dir()):
        cfg = d.name
        try:
            result = json.loads((d / "result.json").read_text())[cfg]
            summary = json.loads((d / f"summary_{cfg}.json").read_text())["kernels"]
        except Exception as err:
            print(f"{cfg}: incomplete ({err})", file=sys.stderr)
            continue
        inv = max(1, int(result.get("timed_invocations", 1)))
        kernels, kernel_us, hbm = {}, 0.0, 0.0
        for name, e in summary.items():
            if not name.startswith("k_"):
                continue
            t, p = e.get("trace", {}), e.get("pmc_avg_per_launch", {})
            per_inv = t.get("calls", 0) / inv
            kernels[name] = {"calls_per_invocation": per_inv, "avg_us": t.get("avg_us"),
                             "us_per_invocation": t.get("total_us", 0.0) / inv,
                             "hbm_bytes_per_launch": p.get("hbm_bytes_per_launch_corrected"),
                             "valu_insts_per_launch": p.get("SQ_INSTS_VALU"), "lds_insts_per_launch": p.get("SQ_INSTS_LDS"),
                             "mfma_f32_mops_per_launch": p.get("SQ_INSTS_VALU_MFMA_MOPS_F32")}
            kernel_us += t.get("total_us", 0.0) / inv
            if p.get("hbm_bytes_per_launch_corrected") is not None:
                hbm += p["hbm_bytes_per_launch_corrected"] * per_inv
        algo = result.get("algorithmic_bytes")
        table[cfg] = {"what": result.get("what"), "wall_ms_per_invocation": result.get("ms"),
                      "kernel_ms_per_invocation_rocprof": kernel_us / 1e3, "algorithmic_bytes": algo,
                      "hbm_bytes_per_invocation_pmc": hbm or None,
                      "traffic_over_algorithmic": (hbm / algo) if algo and hbm else None,
                      "hbm_frac_wall": (algo / (result["ms"] / 1e3) / HBM) if algo and result.get("ms") else None,
                      "hbm_frac_kernels": (algo / (kernel_us / 1e6) / HBM) if algo and kernel_us else None,
                      "audio_seconds_per_s": result.get("audio_seconds_per_s"), "kernels": kernels}
    (root / f"configs_{tag}.json").write_text(json.dumps(table, indent=1))
    lines = ["| config | wall ms | kernel ms (rocprof) | algorithmic MB | PMC traffic / algorithmic | HBM frac (wall) | HBM frac (kernels) | kernels: calls x avg us |",
             "|---|---|---|---|---|---|---|---|"]
    for cfg, e in table.items():
        ks = "; ".join(f"{k} {v['calls_per_invocation']:.0f} x {v['avg_us']:.1f}" for k, v in sorted(e["kernels"].items(), key=lambda kv: -kv[1]["us_per_invocation"]))
        f = lambda v, fmt: (fmt % v) if v is not None else "-"
        lines.append(f"| {cfg} | {f(e['wall_ms_per_invocation'], '%.3f')} | {f(e['kernel_ms_per_invocation_rocprof'], '%.3f')} | "
                     f"{f(e['algorithmic_bytes'] / 1e6 if e['algorithmic_bytes'] else None, '%.1f')} | {f(e['traffic_over_algorithmic'], '%.1f')} | "
                     f"{f(e['hbm_frac_wall'], '%.4f')} | {f(e['hbm_frac_kernels'], '%.4f')} | {ks} |")
    (root / f"configs_{tag}.md").write_text("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
