#!/usr/bin/env python3
"""Per-launch duration series of the step kernel from a rocprofv3 --kernel-trace CSV: quantiles, drift, periodicity."""
import csv, glob, sys
import numpy as np
files = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")
rows = [r for r in csv.DictReader(open(files[0])) if "step_kernel" in r["Kernel_Name"]]
t0 = np.array([int(r["Start_Timestamp"]) for r in rows], np.int64)
t1 = np.array([int(r["End_Timestamp"]) for r in rows], np.int64)
o = np.argsort(t0); t0, t1 = t0[o], t1[o]
d = (t1 - t0) / 1e3
gap = (t0[1:] - t1[:-1]) / 1e3
print(f"{len(d)} launches: duration us  min {d.min():.1f} p10 {np.percentile(d,10):.1f} p50 {np.percentile(d,50):.1f} p90 {np.percentile(d,90):.1f} max {d.max():.1f} mean {d.mean():.1f}")
print(f"gap between launches us: p10 {np.percentile(gap,10):.2f} p50 {np.percentile(gap,50):.2f} p90 {np.percentile(gap,90):.2f} mean {gap.mean():.2f}")
k = len(d) // 10
print("mean duration per tenth of the run:", [round(float(d[i*k:(i+1)*k].mean()), 1) for i in range(10)])
print("slowest launches (index, us):", [(int(i), round(float(d[i]), 1)) for i in np.argsort(d)[-8:]])
print("first 24 durations:", [round(float(x), 1) for x in d[:24]])
mid = d[len(d)//2: len(d)//2 + 24]
print("24 durations mid-run:", [round(float(x), 1) for x in mid])
