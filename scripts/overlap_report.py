#!/usr/bin/env python3
"""How much of the host-to-device copy time of a rocprofv3 `--kernel-trace --memory-copy-trace` run was covered
by kernels running at the same time.  Usage: overlap_report.py <trace dir>  -> JSON on stdout.

Intervals come from the Start/End timestamps (ns) of *_kernel_trace.csv and *_memory_copy_trace.csv.  This
rocprofv3 writes no size column for copies, so ingest copies are told from descriptor / table uploads by their
duration: host-to-device copies of >= 100 us (5.7 MB at the 57 GB/s the link sustains) count."""
import csv
import glob
import json
import os
import sys


def union(intervals):
    out = []
    for a, b in sorted(intervals):
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def intersect_len(u, v):
    i = j = 0
    total = 0
    while i < len(u) and j < len(v):
        lo, hi = max(u[i][0], v[j][0]), min(u[i][1], v[j][1])
        if hi > lo:
            total += hi - lo
        if u[i][1] < v[j][1]:
            i += 1
        else:
            j += 1
    return total


def main():
    root = sys.argv[1]
    kernels, copies, small = [], [], 0
    names = {}
    for path in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            a, b = int(row["Start_Timestamp"]), int(row["End_Timestamp"])
            if "serhip::" in row["Kernel_Name"]:
                kernels.append((a, b))
            key = row["Kernel_Name"].split("(")[0][-40:]
            names[key] = names.get(key, 0) + (b - a)
    for path in glob.glob(os.path.join(root, "**", "*memory_copy_trace.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            direction = row.get("Direction", "")
            if "HOST_TO_DEVICE" not in direction.upper():
                continue
            a, b = int(row["Start_Timestamp"]), int(row["End_Timestamp"])
            if b - a < 100_000:
                small += 1
                continue
            copies.append((a, b))
    uk, uc = union(kernels), union(copies)
    copy_ns = sum(b - a for a, b in uc)
    kern_ns = sum(b - a for a, b in uk)
    both = intersect_len(uk, uc)
    json.dump({
        "h2d_copies_ge_100us": len(copies), "h2d_small_copies": small,
        "h2d_busy_ms": copy_ns / 1e6,
        "serhip_kernel_dispatches": len(kernels), "serhip_kernel_busy_ms": kern_ns / 1e6,
        "copy_and_kernel_concurrent_ms": both / 1e6,
        "fraction_of_serhip_kernel_time_under_a_copy": both / kern_ns if kern_ns else None,
        "fraction_of_copy_time_with_a_serhip_kernel_running": both / copy_ns if copy_ns else None,
        "top_kernels_ms": {k: v / 1e6 for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:8]},
    }, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
