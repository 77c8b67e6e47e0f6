#!/usr/bin/env python3
"""Condense the raw rocprofv3 output of tools/profile_round.sh (gpurun_out/prof_<tag>_{kt,pmc*}) into the tracked summaries:

   profiles/<tag>_kernel_stats.csv   the --kernel-trace --stats table of the bench command (top rows)
   profiles/<round>_counters.json    (round = the tag's prefix, r04_c4 -> r04) per workload: HBM bytes per env step (FETCH_SIZE / WRITE_SIZE), VALU instruction counts,
                                     wave cycles, effective clock - keyed by the kernel-source hash bench.py checks

    python tools/summarize_counters.py <tag> <workload> <kernel substring> <n_envs> <algorithmic read B> <algorithmic B>

Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE come from SEPARATE passes, are in
KiB, WRITE_SIZE is exact, and on gfx950 FETCH_SIZE under-reports streaming reads by 2x (calibrated in round 1 on the no-flow
6-DoF kernel, whose reads are exactly 152 B per env: profiles/r01_pmc_traffic.json "calibration").  SQ_WAVE_CYCLES /
SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles; GRBM_GUI_ACTIVE sums the 8 XCDs.
"""
import csv
import glob
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
OUT = os.path.join(REPO, "gpurun_out")
PROF = os.path.join(REPO, "profiles")


def newest(pattern):
    """gpurun merges the output of several sessions into gpurun_out/: one rocprofv3 run = one <pid> file set; take the latest"""
    files = glob.glob(pattern)
    return [max(files, key=os.path.getmtime)] if files else []


def counters(tag, kernel_substr):
    acc, dur = {}, []
    for d in sorted(glob.glob(os.path.join(OUT, f"prof_{tag}_pmc*"))):
        if not os.path.isdir(d):
            continue
        for f in newest(os.path.join(d, "*", "*_counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if kernel_substr in r["Kernel_Name"]:
                    acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                        dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}, (sum(dur) / len(dur) if dur else None)


def main():
    tag, wl, kernel_substr, n_envs, alg_read, alg_bytes = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), float(sys.argv[5]), float(sys.argv[6])
    from marinevehiclereinforcementlearning_amd import build
    os.makedirs(PROF, exist_ok=True)
    ks = newest(os.path.join(OUT, f"prof_{tag}_kt", "*", "*_kernel_stats.csv"))
    avg_ns = calls = None
    if ks:
        rows = list(csv.reader(open(ks[0])))
        with open(os.path.join(PROF, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
            w = csv.writer(f)
            for r in rows[:8]:
                w.writerow([c[:160] for c in r])
        for r in rows[1:]:
            if kernel_substr in r[0]:
                calls, avg_ns = int(r[1]), float(r[3])
                break
    avg1_ns = calls1 = None
    ks1 = newest(os.path.join(OUT, f"prof_{tag}_kt1", "*", "*_kernel_stats.csv"))
    if ks1:
        rows = list(csv.reader(open(ks1[0])))
        with open(os.path.join(PROF, f"{tag}_chains1_kernel_stats.csv"), "w", newline="") as f:
            w = csv.writer(f)
            for r in rows[:8]:
                w.writerow([c_[:160] for c_ in r])
        for r in rows[1:]:
            if kernel_substr in r[0]:
                calls1, avg1_ns = int(r[1]), float(r[3])
                break
    def medians(which):
        """median / p10 / p90 of the step kernel's dispatch durations (tools/trace_median.py, written before the trace was deleted)"""
        f = os.path.join(OUT, f"prof_{tag}_{which}", "kernel_medians.json")
        if not os.path.exists(f):
            return None
        for k, v in json.load(open(f)).items():
            if kernel_substr in k:
                return {"median_us": v["median_ns"] / 1e3, "p10_us": v["p10_ns"] / 1e3, "p90_us": v["p90_ns"] / 1e3, "calls": v["calls"]}
        return None
    c, cnt, dur_ns = counters(tag, kernel_substr)
    waves = c.get("SQ_WAVES") or (n_envs / 64.0)
    ent = {"tag": tag, "kernel": kernel_substr, "envs": n_envs, "launches_averaged": cnt,
           "bench_command_kernel_avg_us": None if avg_ns is None else avg_ns / 1e3, "bench_command_kernel_calls": calls,
           "bench_command_note": "default bench: 2 chains, each launch covers half the batch and two launches are in flight at any time, "
                                 "so a launch lasts about one whole step",
           "chains1_kernel_avg_us": None if avg1_ns is None else avg1_ns / 1e3, "chains1_kernel_calls": calls1,
           "bench_command_kernel_median": medians("kt"), "chains1_kernel_median": medians("kt1"),
           "median_note": "the --stats average is moved by a few long dispatches (first launches, clock ramps); the MEDIAN is what compares with bench.py's median region",
           "counter_pass_launch": "one launch per env step (--chains 1), 200 steps", "raw": c,
           "algorithmic_read_bytes_per_env": alg_read, "algorithmic_bytes_per_env": alg_bytes}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        rd, wr = 2.0 * c["FETCH_SIZE"] * 1024.0, c["WRITE_SIZE"] * 1024.0
        ent.update({"fetch_correction": 2.0, "hbm_read_bytes_per_step": rd, "hbm_write_bytes_per_step": wr,
                    "hbm_bytes_per_step": rd + wr, "hbm_bytes_per_env": (rd + wr) / n_envs,
                    "hbm_read_bytes_per_env": rd / n_envs, "hbm_write_bytes_per_env": wr / n_envs})
    if "SQ_INSTS_VALU" in c:
        v = {"wave_instr_per_env_step": c["SQ_INSTS_VALU"] / waves,
             "note": "SQ_INSTS_VALU / SQ_WAVES: VALU instructions the wave that owns an env executes per env step"}
        if "SQ_INSTS_VALU_FMA_F32" in c:
            v["flops_per_env_step"] = (2 * c["SQ_INSTS_VALU_FMA_F32"] + c.get("SQ_INSTS_VALU_MUL_F32", 0) + c.get("SQ_INSTS_VALU_ADD_F32", 0)
                                       + c.get("SQ_INSTS_VALU_TRANS_F32", 0)) / waves
        if c.get("SQ_INSTS_VALU_FMA_F64", 0) > c.get("SQ_INSTS_VALU_FMA_F32", 0):   # an fp64 kernel (the fp32 ones form the turbulence sample time in fp64: a handful)
            v["fp64_instr_per_env_step"] = {k_[len("SQ_INSTS_VALU_"):].lower(): c[k_] / waves for k_ in
                                            ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64") if k_ in c}
            v["flops_per_env_step"] = (2 * c["SQ_INSTS_VALU_FMA_F64"] + c.get("SQ_INSTS_VALU_MUL_F64", 0) + c.get("SQ_INSTS_VALU_ADD_F64", 0)
                                       + c.get("SQ_INSTS_VALU_TRANS_F64", 0)) / waves
        if "SQ_WAVE_CYCLES" in c:
            v["wave_cycles_per_wave"] = 4.0 * c["SQ_WAVE_CYCLES"] / waves
            v["valu_active_cycles_per_wave"] = 4.0 * c.get("SQ_ACTIVE_INST_VALU", 0) / waves
            v["wait_inst_cycles_per_wave"] = 4.0 * c.get("SQ_WAIT_INST_ANY", 0) / waves
            v["wait_mem_cycles_per_wave"] = 4.0 * c.get("SQ_WAIT_ANY", 0) / waves
        if "GRBM_GUI_ACTIVE" in c and dur_ns:
            v["effective_clock_GHz_under_profiler"] = c["GRBM_GUI_ACTIVE"] / 8.0 / dur_ns
            v["kernel_us_under_profiler"] = dur_ns / 1e3
        if wl == "c2" and "wave_cycles_per_wave" in v:
            # VERDICT r4 "next 5" (two lanes per env, two thrusters each): what the counters say about it.  C2 is ONE wave per SIMD; a
            # second wave can only use the cycles the first leaves idle, and splitting an env over two lanes makes BOTH run the per-env
            # part of every RK stage (PID, Coriolis / damping / M^-1, kinematics, stage rotation, RK bookkeeping: ~148 of ~220
            # instructions per stage; only allocation rows, saturation / dead-band and jet drag - ~72 - are per thruster)
            busy, life = v["valu_active_cycles_per_wave"], v["wave_cycles_per_wave"]
            share = (220.0 - 36.0 + 6.0) / 220.0        # per lane: half of the per-thruster part saved, 6 instructions of exchange added
            v["two_lanes_per_env_estimate"] = {
                "waves_per_simd_now": 1, "valu_busy_share_of_the_wave_life": busy / life,
                "instructions_per_lane_relative": share, "valu_cycles_of_two_waves_relative_to_the_wave_life_now": 2.0 * share * busy / life,
                "note": "the lone wave keeps its SIMD's VALU busy for %.0f %% of its life (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES); two lanes per env would "
                        "put two waves on the SIMD, each executing %.0f %% of the instructions: by the measured issue rates (profiles/r05_valu_ops.txt: "
                        "2.24 ns per instruction for one wave per SIMD, 1.24 ns for two) 2 x %.2f x 1.24 = %.2f ns of issue per instruction of the present "
                        "kernel against 2.24 now - a 4 %% shorter kernel; not built (DESIGN.md section 5)" % (
                            100.0 * busy / life, 100.0 * share, share, 2.0 * share * 1.24)}
        ent["valu"] = v
    path = os.path.join(PROF, f"{tag.split('_')[0]}_counters.json")
    data = json.load(open(path)) if os.path.exists(path) else {"workloads": {}}
    # identity of the build that was PROFILED: the bench line printed under the profiler carries it
    khash = None
    try:
        khash = json.loads(open(os.path.join(OUT, f"prof_{tag}_kt.json")).read().strip().splitlines()[-1])["roofline"]["kernel_source_hash"]
    except Exception:  # noqa: BLE001
        pass
    if data.get("kernel_source_hash") not in (None, khash):
        data["workloads"] = {}                       # summaries of an older build do not mix with this one
    data["kernel_source_hash"] = khash or build.source_hash()
    try:
        data["commit"] = subprocess.run(["git", "-C", REPO, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except Exception:  # noqa: BLE001
        data["commit"] = None
    data["workloads"][wl] = ent
    json.dump(data, open(path, "w"), indent=1)
    print(json.dumps({k: ent[k] for k in ent if k != "raw"}, indent=1))


if __name__ == "__main__":
    main()
