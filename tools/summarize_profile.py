#!/usr/bin/env python3
"""Condense raw rocprofv3 output (gpurun_out/prof_<tag>_{kt,fetch,write}) into the tracked summaries under profiles/:
   profiles/<tag>_kernel_stats.csv   - the --kernel-trace --stats table (top rows)
   profiles/r01_pmc_traffic.json     - per-launch HBM traffic of the step kernel from the FETCH_SIZE / WRITE_SIZE passes

Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are collected
in SEPARATE passes, are in KiB, WRITE_SIZE is exact, and on gfx950 FETCH_SIZE under-reports streaming reads by 2x.
The 2x factor is calibrated on our own access pattern (dword-per-lane SoA planes): the no-flow 6-DoF kernel reads
exactly 152 B per env (32 state words + 6 action words) - see the "calibration" entry.
"""
import csv
import glob
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "gpurun_out")
PROF = os.path.join(REPO, "profiles")


def counter_avg(tag, which, kernel_substr):
    files = glob.glob(os.path.join(OUT, f"prof_{tag}_{which}", "*", "*_counter_collection.csv"))
    if not files:
        return None, 0
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(files[0])) if kernel_substr in r["Kernel_Name"]]
    return (sum(vals) / len(vals) if vals else None), len(vals)


def main():
    tag, wl, kernel_substr, n_envs, alg_read, alg_bytes = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), float(sys.argv[5]), float(sys.argv[6])
    os.makedirs(PROF, exist_ok=True)
    ks = glob.glob(os.path.join(OUT, f"prof_{tag}_kt", "*", "*_kernel_stats.csv"))
    avg_ns = None
    if ks:
        rows = list(csv.reader(open(ks[0])))
        with open(os.path.join(PROF, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
            w = csv.writer(f)
            for r in rows[:8]:
                w.writerow([c[:160] for c in r])
        for r in rows[1:]:
            if kernel_substr in r[0]:
                avg_ns = float(r[3])
                break
    fetch, nf = counter_avg(tag, "fetch", kernel_substr)
    write, nw = counter_avg(tag, "write", kernel_substr)
    path = os.path.join(PROF, "r01_pmc_traffic.json")
    data = json.load(open(path)) if os.path.exists(path) else {}
    entry = {"tag": tag, "kernel": kernel_substr, "envs": n_envs, "launches_averaged": nf,
             "kernel_avg_us_rocprof": None if avg_ns is None else avg_ns / 1e3,
             "FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB_raw": write,
             "algorithmic_read_bytes_per_env": alg_read, "algorithmic_bytes_per_env": alg_bytes}
    if fetch is not None and write is not None:
        raw_read = fetch * 1024.0
        entry["fetch_raw_bytes_per_env"] = raw_read / n_envs
        entry["fetch_correction"] = 2.0
        entry["hbm_read_bytes_per_launch"] = 2.0 * raw_read
        entry["hbm_write_bytes_per_launch"] = write * 1024.0
        entry["hbm_bytes_per_launch"] = 2.0 * raw_read + write * 1024.0
        entry["hbm_bytes_per_env"] = entry["hbm_bytes_per_launch"] / n_envs
    data[wl] = entry
    json.dump(data, open(path, "w"), indent=1)
    print(json.dumps(entry, indent=1))


if __name__ == "__main__":
    main()
