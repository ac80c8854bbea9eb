#!/usr/bin/env python3
"""Copy the small artefacts of one `scripts/profile_gpu.sh <tag>` run (merged back under gpurun_out/)
into profiles/: trimmed kernel stats (this library's kernels only), the per-kernel summary, the bench
line, and profiles/traffic.json (what bench.py reports as roofline.traffic).

Usage: python scripts/collect_profile.py <tag>
"""
import csv
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CLIPS = 12500
ALGO = 192748


def main():
    tag = sys.argv[1]
    src = ROOT / "gpurun_out" / f"prof_{tag}"
    dst = ROOT / "profiles"
    shutil.copy(src / f"summary_{tag}.json", dst / f"{tag}_summary.json")
    rows = list(csv.reader(open(src / f"kernel_stats_{tag}.csv")))
    with open(dst / f"{tag}_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(rows[0])
        for r in rows[1:]:
            if "serhip::" in r[0]:
                w.writerow(r)
    keep, on = [], False
    for line in (src / "summary.txt").read_text().split("\n"):
        if not line.startswith(" "):
            on = line.startswith(("k_stft", "k_project", "k_segment", "k_tuning", "k_dft"))
        if on:
            keep.append(line)
    (dst / f"{tag}_summary.txt").write_text(
        f"# rocprofv3 --kernel-trace --stats + PMC passes of `python3 bench.py --steps 50 --warmup 5 --cpu-clips 0` "
        f"(scripts/profile_gpu.sh {tag})\n# per-launch averages; one launch = 12 500 clips = 1 175 000 STFT frames\n"
        + "\n".join(keep) + "\n")
    bench = ROOT / "gpurun_out" / f"bench_{tag}.json"
    if bench.exists():
        shutil.copy(bench, dst / f"{tag}_bench.json")
    in_trace = src / f"bench_in_trace_{tag}.json"
    if in_trace.exists() and in_trace.stat().st_size:
        shutil.copy(in_trace, dst / f"{tag}_bench_in_trace.json")
    kernels = json.load(open(dst / f"{tag}_summary.json"))["kernels"]
    k = kernels["k_stft_frames"]["pmc_avg_per_launch"]
    json.dump({
        "tag": tag,
        "k_stft_frames_rocprof_avg_us": kernels["k_stft_frames"]["trace"]["avg_us"],
        "k_stft_frames_rocprof_median_us": kernels["k_stft_frames"]["trace"]["median_us"],
        "k_stft_frames_rocprof_launches": kernels["k_stft_frames"]["trace"]["calls"],
        "k_project_frames_rocprof_avg_us": kernels.get("k_project_frames", {}).get("trace", {}).get("avg_us"),
        "source": f"profiles/{tag}_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                  "FETCH_SIZE doubled per MI355X_MICROARCH.md)",
        "k_stft_frames_hbm_bytes_per_launch": k["hbm_bytes_per_launch_corrected"],
        "k_stft_frames_hbm_bytes_per_launch_raw": k["hbm_bytes_per_launch_raw"],
        "k_stft_frames_segments_per_launch_avg": CLIPS,
        "k_stft_frames_algorithmic_bytes_per_launch_avg": CLIPS * ALGO,
    }, open(dst / "traffic.json", "w"), indent=1)
    for name, e in sorted(kernels.items()):
        if not name.startswith("k_"):
            continue
        p = e.get("pmc_avg_per_launch", {})
        print(f"{name:22s} avg {e['trace']['avg_us']:8.1f} us  VALU/frame {p.get('SQ_INSTS_VALU', 0) / 1.175e6:6.1f}  "
              f"LDS/frame {p.get('SQ_INSTS_LDS', 0) / 1.175e6:5.1f}  HBM {p.get('hbm_bytes_per_launch_corrected', 0) / 1e9:5.2f} GB")


if __name__ == "__main__":
    main()
