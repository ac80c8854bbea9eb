#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace + PMC passes) into small per-kernel summaries.

Usage: summarize_profile.py <prof_dir> <tag>   -> writes <prof_dir>/summary_<tag>.json and prints a table.
FETCH_SIZE is doubled per /opt/skills/guides/MI355X_MICROARCH.md (gfx950 reports half of a wide
coalesced read stream); both raw and corrected values are kept.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"k_segment_finish<(\d)>", name) or re.search(r"k_segment_finishILi(\d)E", name)
    if m:
        return f"k_segment_finish<{m.group(1)}>"
    for key in ("k_prepare_convert", "k_prepare_clip"):
        m = re.search(key + r"<(\d)>", name) or re.search(key + r"ILi(\d)E", name)
        if m:
            return f"{key}<{m.group(1)}>"
    for key in ("k_project_frames", "k_stft_frames", "k_tuning_waves", "k_segment_finish", "k_dft_frames", "k_pool_windows", "k_whisper_frames",
                "k_whisper_finalize", "k_prepare_normalise", "k_prepare_convert", "k_prepare_clip", "k_mlp_scale", "k_mlp_dense", "k_mlp_output"):
        if key in name:
            return key
    return name[:60]


def main():
    prof, tag = sys.argv[1], sys.argv[2]
    summary = {"tag": tag, "kernels": {}}
    # kernel trace: per-dispatch durations
    dur = defaultdict(list)
    for path in glob.glob(os.path.join(prof, "trace", "**", "*kernel_trace.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                dur[short(row["Kernel_Name"])].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    for k, v in dur.items():
        v_sorted = sorted(v)
        summary["kernels"].setdefault(k, {})["trace"] = {
            "calls": len(v), "total_us": sum(v), "avg_us": sum(v) / len(v), "min_us": v_sorted[0],
            "median_us": v_sorted[len(v) // 2], "max_us": v_sorted[-1]}
    # PMC passes: per-dispatch counter values averaged per kernel
    pmc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(prof, "pmc*", "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                pmc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, counters in pmc.items():
        entry = summary["kernels"].setdefault(k, {}).setdefault("pmc_avg_per_launch", {})
        for c, vals in counters.items():
            entry[c] = sum(vals) / len(vals)
        if "FETCH_SIZE" in entry or "WRITE_SIZE" in entry:
            fetch_kb = entry.get("FETCH_SIZE", 0.0)
            write_kb = entry.get("WRITE_SIZE", 0.0)
            entry["hbm_bytes_per_launch_raw"] = (fetch_kb + write_kb) * 1024.0
            entry["hbm_bytes_per_launch_corrected"] = (2.0 * fetch_kb + write_kb) * 1024.0
    out = os.path.join(prof, f"summary_{tag}.json")
    with open(out, "w") as f:
        json.dump(summary, f, indent=1, sort_keys=True)
    for k, e in sorted(summary["kernels"].items()):
        t = e.get("trace", {})
        print(f"{k:22s} calls={t.get('calls', 0):4d} avg_us={t.get('avg_us', 0):10.1f} total_us={t.get('total_us', 0):12.1f}")
        for c, v in sorted(e.get("pmc_avg_per_launch", {}).items()):
            print(f"    {c:36s} {v:18.1f}")


if __name__ == "__main__":
    main()
