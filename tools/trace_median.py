#!/usr/bin/env python3
"""Per-kernel median / p10 / p90 / mean of the dispatch durations in a rocprofv3 --kernel-trace CSV (the --stats table only has
the mean, which a handful of long outliers moves by 20 %: profiles/r04_auv4m_*).  Writes <dir>/kernel_medians.json next to the
trace; tools/summarize_counters.py puts the figures beside the averages.      python tools/trace_median.py <rocprof output dir>"""
import csv
import glob
import json
import os
import sys


def main(d):
    out = {}
    for f in glob.glob(os.path.join(d, "*", "*_kernel_trace.csv")):
        by = {}
        for r in csv.DictReader(open(f)):
            by.setdefault(r["Kernel_Name"], []).append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        for k, v in by.items():
            v.sort()
            n = len(v)
            out[k[:200]] = {"calls": n, "median_ns": v[n // 2], "p10_ns": v[n // 10], "p90_ns": v[(9 * n) // 10], "mean_ns": sum(v) / n}
    with open(os.path.join(d, "kernel_medians.json"), "w") as f:
        json.dump(out, f, indent=1)
    return out


if __name__ == "__main__":
    main(sys.argv[1])
